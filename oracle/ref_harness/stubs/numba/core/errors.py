class TypingError(Exception):
    pass
