"""Logging hook with the reference's singleton shape (tunempc/logger.py:30-50): `Logger.logger` is a
stdlib logger named 'tunempc'; the hot path emits the same INFO banners as convexifier.py."""
import logging


class Logger:
    logger = logging.getLogger('tunempc')
