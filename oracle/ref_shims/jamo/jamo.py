import unicodedata

_BASE, _LEAD0, _VOW0, _TAIL0 = 0xAC00, 0x1100, 0x1161, 0x11A7


def _decompose(ch):
    cp = ord(ch)
    if 0xAC00 <= cp <= 0xD7A3:
        r = cp - _BASE
        out = [chr(_LEAD0 + r // 588), chr(_VOW0 + (r % 588) // 28)]
        if r % 28:
            out.append(chr(_TAIL0 + r % 28))
        return out
    return [ch]


def hangul_to_jamo(s):
    return (j for ch in s for j in _decompose(ch))


def h2j(s):
    return ''.join(hangul_to_jamo(s))


def is_hcj(c):
    cp = ord(c)
    return 0x3131 <= cp <= 0x318E and cp != 0x3164


def hcj_to_jamo(c, position='vowel'):
    name = unicodedata.name(c, '')
    if not name.startswith('HANGUL LETTER '):
        return c
    tag = {'lead': 'CHOSEONG', 'vowel': 'JUNGSEONG', 'tail': 'JONGSEONG'}[position]
    try:
        return unicodedata.lookup('HANGUL %s %s' % (tag, name[len('HANGUL LETTER '):]))
    except KeyError:
        return c


def _jamo_char_to_hcj(c):
    name = unicodedata.name(c, '')
    for tag in ('CHOSEONG', 'JUNGSEONG', 'JONGSEONG'):
        p = 'HANGUL %s ' % tag
        if name.startswith(p):
            try:
                return unicodedata.lookup('HANGUL LETTER ' + name[len(p):])
            except KeyError:
                return c
    return c


def j2h(lead, vowel, tail=None):
    t = (ord(tail) - _TAIL0) if tail else 0
    return chr(_BASE + (ord(lead) - _LEAD0) * 588 + (ord(vowel) - _VOW0) * 28 + t)
