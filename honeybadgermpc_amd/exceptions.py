"""Exception hierarchy (reference: honeybadgermpc/exceptions.py)."""


class HoneyBadgerMPCError(Exception):
    """Base class of the errors raised by this package's protocol layer."""
