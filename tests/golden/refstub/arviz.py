"""Empty stand-in: gwinferno/preprocess/data_collection.py imports arviz at module level (:7) but the
functions the golden generator calls never touch it.  Build container only."""
