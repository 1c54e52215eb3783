"""WESTPA plugin level of the drop-in boundary."""
from .hamsm_driver import HAMSMDriver  # noqa: F401
