"""Shim for jamo 0.4.1 — Unicode arithmetic restatement of the calls used by
text/korean.py:7-8,57-63,182-183."""
from .jamo import (hangul_to_jamo, h2j, j2h, hcj_to_jamo, is_hcj,  # noqa
                   _jamo_char_to_hcj)
