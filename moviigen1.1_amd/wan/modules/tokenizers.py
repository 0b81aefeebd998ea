"""Prompt -> token ids in front of the umT5 encoder: the `HuggingfaceTokenizer` the reference builds in
T5EncoderModel (wan/modules/tokenizers.py:50-82; used at t5.py:500-513 with clean='whitespace').

Same constructor and call contract (`tok(texts, return_mask=True, add_special_tokens=True)` ->
(ids [B, seq_len], mask [B, seq_len])); the text cleaning is a small table of named pipelines.

`ftfy.fix_text` (reference tokenizers.py:13) is used when it is installed.  This image does not have
it, so its DEFAULT fixes that change well-formed prompts are restated below (`_fix_text_defaults`):
terminal escapes removed, latin ligatures expanded, fullwidth / halfwidth forms folded (the default
negative prompt `config.sample_neg_prompt` is full of U+FF0C fullwidth commas -> ','), curly quotes
straightened, line breaks unified, control characters removed, NFC.  The one thing not restated is the
mojibake (mis-decoded UTF-8) repair: text that looks like mojibake logs a warning instead."""
import html
import logging
import string
import unicodedata

import regex

__all__ = ['HuggingfaceTokenizer']

# ftfy.fixes: fix_latin_ligatures / fix_character_width / uncurl_quotes / fix_line_breaks / remove_control_chars /
# remove_terminal_escapes, restated from their documented behaviour (code points written as escapes on purpose)
_LIGATURES = {
    0x0132: 'IJ', 0x0133: 'ij', 0x0149: 'ʼn', 0x01f1: 'DZ', 0x01f2: 'Dz', 0x01f3: 'dz',
    0x01c4: 'DŽ', 0x01c5: 'Dž', 0x01c6: 'dž', 0x01c7: 'LJ', 0x01c8: 'Lj', 0x01c9: 'lj',
    0x01ca: 'NJ', 0x01cb: 'Nj', 0x01cc: 'nj', 0xfb00: 'ff', 0xfb01: 'fi', 0xfb02: 'fl', 0xfb03: 'ffi',
    0xfb04: 'ffl', 0xfb05: 'ſt', 0xfb06: 'st'}
_WIDTH = {0x3000: ' '}
_WIDTH.update({c: unicodedata.normalize('NFKC', chr(c)) for c in range(0xFF01, 0xFFF0)})
_CONTROL = {c: None for c in (list(range(0x00, 0x09)) + [0x0b] + list(range(0x0e, 0x20)) + [0x7f] +
                              list(range(0x206a, 0x2070)) + [0xfeff] + list(range(0xfff9, 0xfffd)))}
_ANSI = regex.compile('\x1b\\[((?:\\d|;)*)([a-zA-Z])')
_SINGLE_Q = regex.compile('[ʼ‘-‛]')
_DOUBLE_Q = regex.compile('[“-‟]')
_LINE_BREAKS = regex.compile('\r\n|[\r  ]')
_MOJIBAKE_HINT = regex.compile('[ÂÃâ][-¿‘-›€]')
_warned = set()


def _warn_once(key, msg):
    if key not in _warned:
        _warned.add(key)
        logging.warning(msg)


def _fix_text_defaults(text):
    _warn_once('ftfy', 'ftfy is not installed: prompt cleaning uses the restated ftfy.fix_text defaults (width folding, '
                       'quotes, ligatures, line breaks, control characters, NFC); mojibake repair is unavailable')
    if _MOJIBAKE_HINT.search(text):
        _warn_once('mojibake', 'prompt looks like mis-decoded UTF-8 (mojibake); install ftfy to get the reference repair')
    text = _ANSI.sub('', text)
    text = text.translate(_LIGATURES).translate(_WIDTH)
    text = _DOUBLE_Q.sub('"', _SINGLE_Q.sub("'", text))
    text = _LINE_BREAKS.sub('\n', text)
    text = text.translate(_CONTROL)
    return unicodedata.normalize('NFC', text)


try:
    from ftfy import fix_text as _repair
except ModuleNotFoundError:
    _repair = _fix_text_defaults

_SPACES = regex.compile(r'\s+')
_NO_PUNCT = str.maketrans('', '', string.punctuation)


def _unescape_twice(text):
    return html.unescape(html.unescape(_repair(text))).strip()


def _squeeze(text):
    return _SPACES.sub(' ', text).strip()


def _canonical(text):
    return _squeeze(text.replace('_', ' ').translate(_NO_PUNCT).lower())


# cleaning mode -> pipeline applied left to right
_CLEANERS = {
    None: (),
    'whitespace': (_unescape_twice, _squeeze),
    'lower': (_unescape_twice, _squeeze, str.lower),
    'canonicalize': (_unescape_twice, _canonical),
}


def clean_text(text, mode):
    """the text the tokenizer sees for cleaning mode `mode` (reference tokenizers.py:75-82)."""
    for step in _CLEANERS[mode]:
        text = step(text)
    return text


class HuggingfaceTokenizer:

    def __init__(self, name, seq_len=None, clean=None, **kwargs):
        if clean not in _CLEANERS:
            raise AssertionError(f'clean must be one of {sorted(k for k in _CLEANERS if k)} or None')
        from transformers import AutoTokenizer
        self.name, self.seq_len, self.clean = name, seq_len, clean
        self.tokenizer = AutoTokenizer.from_pretrained(name, **kwargs)
        self.vocab_size = self.tokenizer.vocab_size

    def _prepare(self, text):
        return clean_text(text, self.clean)

    def __call__(self, sequence, **kwargs):
        want_mask = kwargs.pop('return_mask', False)
        texts = [sequence] if isinstance(sequence, str) else list(sequence)
        call = dict(return_tensors='pt')
        if self.seq_len is not None:      # fixed-length batches: pad and cut to seq_len (512 for the T2V path)
            call.update(padding='max_length', truncation=True, max_length=self.seq_len)
        call.update(kwargs)
        enc = self.tokenizer([self._prepare(t) for t in texts], **call)
        return (enc.input_ids, enc.attention_mask) if want_mask else enc.input_ids
