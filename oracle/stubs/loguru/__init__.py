import logging

logger = logging.getLogger("loguru-stub")
