"""Korean grapheme front end: text → jamo symbols → ids (integer path, bit-exact by contract).

Own implementation of the behaviour of the reference's text/korean.py (`tokenize` 177-195,
`normalize` 233-249, `number_to_korean` 322-394, symbol table `ALL_SYMBOLS_1` korean.py:24) without
the `jamo`/`nltk` packages: Hangul syllables are decomposed arithmetically
(cp-0xAC00 = 588·lead + 28·vowel + tail).  Quirks kept on purpose: the
80-entry table lists 'ㅇ' twice, so tail ᆮ and tail ᆼ both encode to id 62 and id 48 is never
produced.
"""
import ast
import re

PAD, EOS = '_', '~'
PUNC = "!'(),-.:;?"
SPACE = ' '

_N_LEAD, _N_VOWEL, _N_TAIL = 19, 21, 27
_LEADS = [chr(0x1100 + i) for i in range(_N_LEAD)]
_VOWELS_COMPAT = [chr(0x314F + i) for i in range(_N_VOWEL)]
# tails ᆨ..ᇂ as the reference spells them; slot 6 (tail ᆮ) carries 'ㅇ' (the table's duplicate)
_TAILS_COMPAT = list("ㄱㄲㄳㄴㄵㄶㅇㄹㄺㄻㄼㄽㄾㄿㅀㅁㅂㅄㅅㅆㅇㅈㅊㅋㅌㅍㅎ")
assert len(_TAILS_COMPAT) == _N_TAIL

SYMBOLS = [PAD, EOS] + _LEADS + _VOWELS_COMPAT + _TAILS_COMPAT + list(PUNC) + [SPACE]
assert len(SYMBOLS) == 80
SYMBOL_TO_ID = {s: i for i, s in enumerate(SYMBOLS)}       # later duplicate wins → 'ㅇ' = 62

# conjoining jamo → table symbol
_JAMO_TO_SYMBOL = {PAD: PAD, EOS: EOS}
_JAMO_TO_SYMBOL.update({c: c for c in _LEADS})
_JAMO_TO_SYMBOL.update({chr(0x1161 + i): _VOWELS_COMPAT[i] for i in range(_N_VOWEL)})
_JAMO_TO_SYMBOL.update({chr(0x11A8 + i): _TAILS_COMPAT[i] for i in range(_N_TAIL)})
_JAMO_TO_SYMBOL.update({c: c for c in PUNC + SPACE})

# stand-alone compatibility consonants (U+3131…) → lead jamo when such a lead exists
_COMPAT_LEADS = dict(zip("ㄱㄲㄴㄷㄸㄹㅁㅂㅃㅅㅆㅇㅈㅉㅊㅋㅌㅍㅎ", _LEADS))


class TextFrontendError(SystemExit):
    """The reference swallows every cleaning error and calls exit() (text/__init__.py:55-57);
    this is a SystemExit too, but it says what went wrong."""


def decompose(text):
    """Hangul syllables → conjoining jamo; everything else passes through."""
    out = []
    for ch in text:
        cp = ord(ch)
        if 0xAC00 <= cp <= 0xD7A3:
            r = cp - 0xAC00
            out.append(chr(0x1100 + r // 588))
            out.append(chr(0x1161 + (r % 588) // 28))
            if r % 28:
                out.append(chr(0x11A7 + r % 28))
        else:
            out.append(ch)
    return out


# ----------------------------------------------------------------------------- normalisation
# Small built-in lexicon (whole-word English → Hangul reading, and literal replacements).  The
# reference ships a corpus-specific table (text/ko_dictionary.py); extend with register_words().
_LITERALS = {'2 30대': '이삼십대', '20~30대': '이삼십대', '20, 30대': '이십대 삼십대', '1+1': '원플러스원',
             '3에서 6개월인': '3개월에서 육개월인', 'mp3': '엠피쓰리'}
_LETTER = dict(A='에이', B='비', C='씨', D='디', E='이', F='에프', G='지', H='에이치', I='아이', J='제이',
               K='케이', L='엘', M='엠', N='엔', O='오', P='피', Q='큐', R='알', S='에스', T='티', U='유',
               V='브이', W='더블유', X='엑스', Y='와이', Z='지')
_WORDS = {'DVD': '디비디', 'CCTV': '씨씨티비', 'TV': '티비', 'GV': '지비', 'KOREA': '코리아', 'idol': '아이돌',
          'track': '트랙', 'down': '다운', 'up': '업', 'bill': '빌', 'OO': '오오'}
for _c in 'PLTBCOYS':
    _WORDS[_c] = _LETTER[_c]
    _WORDS[_c.lower()] = _LETTER[_c]
_WORDS['X'] = _LETTER['X']

_UNITS_1 = {'%': '퍼센트', 'cm': '센치미터', 'mm': '밀리미터', 'km': '킬로미터', 'kg': '킬로그람'}
_UNITS_2 = {'m': '미터'}
_DIGIT = dict(zip('0123456789', '영일이삼사오육칠팔구'))
_SINO = [""] + list("일이삼사오육칠팔구")
_MYRIAD = [""] + list("만억조경해")
_POS = [""] + list("십백천")
_NATIVE = [""] + ["한", "두", "세", "네", "다섯", "여섯", "일곱", "여덟", "아홉"]
_NATIVE_TENS = {"십": "열", "두십": "스물", "세십": "서른", "네십": "마흔", "다섯십": "쉰", "여섯십": "예순",
                "일곱십": "일흔", "여덟십": "여든", "아홉십": "아흔"}
_NUMBER = r"([+-]?\d[\d,]*)[\.]?\d*"
_COUNTER = "(시|명|가지|살|마리|포기|송이|수|톨|통|점|개|벌|척|채|다발|그루|자루|줄|켤레|그릇|잔|마디|상자|사람|곡|병|판)"
_QUOTED = """([`"'＂“‘])(.+?)([`"'＂”’])"""
_HANJA_PAREN = r'\([⺀-⺙⺛-⻳⼀-⿕々〇〡-〩〸-〺〻㐀-䶵一-鿃豈-鶴侮-頻並-龎]+\)'


def register_words(words=None, literals=None):
    """Extend the lexicon (e.g. with a corpus-specific English→Hangul table)."""
    _WORDS.update(words or {})
    _LITERALS.update(literals or {})


def _replace_literals(text, table):
    if not any(k in text for k in table):
        return text
    pat = re.compile('|'.join(re.escape(k) for k in table))
    return pat.sub(lambda m: table[m.group()], text)


def read_number(num_str, counter=None):
    """Digits → Hangul reading.  counter != None selects native-Korean counting words."""
    is_count = counter is not None
    num_str = num_str.replace(',', '')
    try:
        num = ast.literal_eval(num_str)
    except Exception:
        num = int(num_str)
    if num == 0:
        return "영"
    parts = num_str.split('.')
    if len(parts) > 2:
        raise ValueError("wrong number format: %r" % num_str)
    digit_str, frac = parts[0], (parts[1] if len(parts) == 2 else None)
    if is_count and frac is not None:
        raise ValueError("counter words cannot take a fractional number")
    digit = int(digit_str)
    if digit_str.startswith("-"):
        digit, digit_str = abs(digit), str(abs(digit))
    size = len(str(digit))
    kor, group = "", []
    for i, ch in enumerate(digit_str, start=1):   # NB: digit_str may carry a sign / leading zeros
        v = int(ch)
        if v != 0:
            group += (_NATIVE if is_count else _SINO)[v]
            group += _POS[(size - i) % 4]
        if (size - i) % 4 == 0 and len(group) != 0:
            kor += "".join(group)
            group = []
            kor += _MYRIAD[int((size - i) / 4)]
    if is_count:
        if kor.startswith("한") and len(kor) > 1:
            kor = kor[1:]
        if any(w in kor for w in _NATIVE_TENS):
            kor = re.sub('|'.join(_NATIVE_TENS.keys()), lambda m: _NATIVE_TENS[m.group()], kor)
    if not is_count and kor.startswith("일") and len(kor) > 1:
        kor = kor[1:]
    if frac is not None:
        kor += "쩜 " + re.sub(r'\d', lambda m: _DIGIT[m.group()], frac)
    if num_str.startswith("+"):
        kor = "플러스 " + kor
    elif num_str.startswith("-"):
        kor = "마이너스 " + kor
    return kor + (counter or "")


def _split_sentences(s):
    parts = re.split(r'(?<=[.!?])\s+', s.strip())
    return [p for p in parts if p]


def normalize(text):
    text = text.strip().replace("'", "").replace('"', "")
    text = re.sub(r'\(\d+일\)', '', text)
    text = re.sub(_HANJA_PAREN, '', text)
    text = _replace_literals(text, _LITERALS)
    text = re.sub("([A-Za-z]+)", lambda m: _WORDS.get(m.group(), m.group()), text)
    text = re.sub('[a-zA-Z]+', lambda m: "".join(_LETTER[c] for c in m.group(0))
                  if all(c.isupper() for c in m.group(0)) else m.group(0), text)
    text = re.sub(_QUOTED, lambda m: " ".join("'{}'".format(x) for x in _split_sentences(m.group()[1:-1])), text)
    text = _replace_literals(text, _UNITS_1)
    text = _replace_literals(text, _UNITS_2)
    text = re.sub(_NUMBER + _COUNTER, lambda m: read_number(m.group(1), m.group(2)), text)
    text = re.sub(_NUMBER, lambda m: read_number(m.group()), text)
    return text


def tokenize(text, as_id=False):
    """normalise → jamo → table symbols (+ EOS), like reference tokenize(symbol_type=1)."""
    tokens = []
    for tok in decompose(normalize(text)):
        tok = _COMPAT_LEADS.get(tok, tok)
        try:
            tokens.append(_JAMO_TO_SYMBOL[tok])
        except KeyError:
            raise TextFrontendError("symbol %r (U+%04X) is not in the Korean symbol table" % (tok, ord(tok)))
    tokens.append(EOS)
    return [SYMBOL_TO_ID[t] for t in tokens] if as_id else tokens
