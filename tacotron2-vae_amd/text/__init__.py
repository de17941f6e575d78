"""text → id sequence (reference text/__init__.py:30-60 with `korean_cleaners`, cleaners.py:93-97)."""
import re

from . import korean

_curly_re = re.compile(r'(.*?)\{(.+?)\}(.*)')
symbols = korean.SYMBOLS
_symbol_to_id = korean.SYMBOL_TO_ID
_id_to_symbol = {i: s for i, s in enumerate(symbols)}


def korean_cleaners(text):
    return korean.tokenize(text, as_id=False)


_CLEANERS = {'korean_cleaners': korean_cleaners}


def _clean_text(text, cleaner_names):
    for name in cleaner_names:
        if name not in _CLEANERS:
            raise Exception('Unknown cleaner: %s (only the Korean path is on the MI355X hot path)' % name)
        text = _CLEANERS[name](text)
    return text


def _symbols_to_sequence(tokens):
    return [_symbol_to_id[s] for s in tokens if s in _symbol_to_id and s != '_' and s != '~']


def text_to_sequence(text, cleaner_names=('korean_cleaners',)):
    """Ids in [1,79]; exactly one EOS (id 1) appended.  `{...}` ARPAbet spans are dropped, which is
    what the reference's Korean table does to them (no '@' symbols in it)."""
    sequence = []
    while len(text):
        m = _curly_re.match(text)
        if not m:
            sequence += _symbols_to_sequence(_clean_text(text, cleaner_names))
            break
        sequence += _symbols_to_sequence(_clean_text(m.group(1), cleaner_names))
        text = m.group(3)
    sequence.append(_symbol_to_id['~'])
    return sequence


def sequence_to_text(sequence):
    return ''.join(_id_to_symbol[i] for i in sequence if i in _id_to_symbol)
