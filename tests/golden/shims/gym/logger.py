_LEVEL = 30


def set_level(level):
    global _LEVEL
    _LEVEL = level


def warn(msg, *args):
    pass
