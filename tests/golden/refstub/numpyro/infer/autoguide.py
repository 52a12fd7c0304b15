class AutoDelta:
    pass
