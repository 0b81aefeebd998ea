"""HuggingfaceTokenizer — drop-in for reference wan/modules/tokenizers.py:50-82 (the prompt ->
token ids step in front of the umT5 encoder).  `ftfy` (mojibake repair) is used when installed;
without it the text passes through NFC normalisation only — the ids then still match for
well-formed input."""
import html
import string
import unicodedata

import regex as re

__all__ = ['HuggingfaceTokenizer']

try:
    import ftfy
    _fix = ftfy.fix_text
except ModuleNotFoundError:                     # not in this image; see module docstring
    def _fix(text):
        return unicodedata.normalize('NFC', text)


def basic_clean(text):
    return html.unescape(html.unescape(_fix(text))).strip()


def whitespace_clean(text):
    return re.sub(r'\s+', ' ', text).strip()


def canonicalize(text, keep_punctuation_exact_string=None):
    text = text.replace('_', ' ')
    table = str.maketrans('', '', string.punctuation)
    if keep_punctuation_exact_string:
        text = keep_punctuation_exact_string.join(p.translate(table) for p in text.split(keep_punctuation_exact_string))
    else:
        text = text.translate(table)
    return re.sub(r'\s+', ' ', text.lower()).strip()


class HuggingfaceTokenizer:

    def __init__(self, name, seq_len=None, clean=None, **kwargs):
        assert clean in (None, 'whitespace', 'lower', 'canonicalize')
        from transformers import AutoTokenizer
        self.name, self.seq_len, self.clean = name, seq_len, clean
        self.tokenizer = AutoTokenizer.from_pretrained(name, **kwargs)
        self.vocab_size = self.tokenizer.vocab_size

    def __call__(self, sequence, **kwargs):
        return_mask = kwargs.pop('return_mask', False)
        opts = {'return_tensors': 'pt'}
        if self.seq_len is not None:
            opts.update(padding='max_length', truncation=True, max_length=self.seq_len)
        opts.update(**kwargs)
        if isinstance(sequence, str):
            sequence = [sequence]
        if self.clean:
            sequence = [self._clean(u) for u in sequence]
        enc = self.tokenizer(sequence, **opts)
        return (enc.input_ids, enc.attention_mask) if return_mask else enc.input_ids

    def _clean(self, text):
        if self.clean == 'whitespace':
            return whitespace_clean(basic_clean(text))
        if self.clean == 'lower':
            return whitespace_clean(basic_clean(text)).lower()
        return canonicalize(basic_clean(text))
