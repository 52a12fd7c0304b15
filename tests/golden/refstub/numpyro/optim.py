class Adam:
    pass
