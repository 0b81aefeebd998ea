"""Prompt -> token ids in front of the umT5 encoder: the `HuggingfaceTokenizer` the reference builds in
T5EncoderModel (wan/modules/tokenizers.py:50-82; used at t5.py:500-513 with clean='whitespace').

Same constructor and call contract (`tok(texts, return_mask=True, add_special_tokens=True)` ->
(ids [B, seq_len], mask [B, seq_len])); the text cleaning is a small table of named pipelines.
`ftfy` (mojibake repair) is applied when it is installed; this image does not have it, and for
well-formed prompts NFC normalisation gives the same ids."""
import html
import string
import unicodedata

import regex

__all__ = ['HuggingfaceTokenizer']

try:
    from ftfy import fix_text as _repair
except ModuleNotFoundError:
    def _repair(text):
        return unicodedata.normalize('NFC', text)

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


class HuggingfaceTokenizer:

    def __init__(self, name, seq_len=None, clean=None, **kwargs):
        if clean not in _CLEANERS:
            raise AssertionError(f'clean must be one of {sorted(k for k in _CLEANERS if k)} or None')
        from transformers import AutoTokenizer
        self.name, self.seq_len, self.clean = name, seq_len, clean
        self.tokenizer = AutoTokenizer.from_pretrained(name, **kwargs)
        self.vocab_size = self.tokenizer.vocab_size

    def _prepare(self, text):
        for step in _CLEANERS[self.clean]:
            text = step(text)
        return text

    def __call__(self, sequence, **kwargs):
        want_mask = kwargs.pop('return_mask', False)
        texts = [sequence] if isinstance(sequence, str) else list(sequence)
        call = dict(return_tensors='pt')
        if self.seq_len is not None:      # fixed-length batches: pad and cut to seq_len (512 for the T2V path)
            call.update(padding='max_length', truncation=True, max_length=self.seq_len)
        call.update(kwargs)
        enc = self.tokenizer([self._prepare(t) for t in texts], **call)
        return (enc.input_ids, enc.attention_mask) if want_mask else enc.input_ids
