def unidecode(s):  # English path unused by korean_cleaners (cleaners.py:16)
    return s
