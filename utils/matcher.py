"""`utils.matcher` of the reference (/root/reference/utils/matcher.py:93-133) -> walkgpt_amd.matcher."""
from walkgpt_amd.matcher import match_pred  # noqa: F401
