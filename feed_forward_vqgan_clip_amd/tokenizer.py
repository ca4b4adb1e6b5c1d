"""CLIP byte-pair tokenizer front-end (`clip.tokenize`, used by the reference at main.py:266,345,418,1034,1302,1305).

The algorithm is the published CLIP `SimpleTokenizer` [clip-anytorch 2.2.0, upstream, not in /root/reference]:
lower-cased, whitespace-cleaned text -> regex pre-tokens -> bytes mapped to printable unicode -> greedy BPE merges by
rank -> ids, wrapped as `[SOT] + ids + [EOT]`, zero padded to the context length, `truncate=True` forcing the last
position to EOT.  Output is int64 because main.py:733 only treats `torch.long` inputs as tokens.

No vocabulary ships with this repo (no network): pass `bpe_path` or set `FFVC_BPE_VOCAB` to CLIP's
`bpe_simple_vocab_16e6.txt.gz` (plain `.txt` also accepted).  Without it `tokenize` raises.

Pre-token pattern: CLIP's own, with the unicode classes `\\p{L}` (letters) and `\\p{N}` (numbers) through the `regex` module,
so accented / Cyrillic / CJK text splits exactly as upstream (ids are integer work: bit-exact or wrong).
Cleaning: upstream runs `ftfy.fix_text` first.  `ftfy` is used when it is importable; otherwise `_fix_text_basic` applies the
deterministic, text-local part of ftfy's default configuration (NFC normalisation, curly quotes -> straight, Latin ligatures,
full-width -> half-width characters, line-break and control-character clean-up, C1 controls -> their Windows-1252 characters).
NOT reproduced without ftfy: its heuristic mojibake repair (`fix_encoding`, e.g. "Ã©" -> "é") and lossy-sequence
restoration — prompts that contain encoding damage tokenise as written.
"""
import gzip
import html
import os
import re
import unicodedata
from functools import lru_cache

import regex
import torch

try:                                   # not in the offline image; used when present so the ids match upstream on damaged text too
    import ftfy as _ftfy
except ImportError:                    # pragma: no cover - depends on the environment
    _ftfy = None

_PAT = regex.compile(r"""<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+""",
                     regex.IGNORECASE)

_QUOTES = {0x2018: "'", 0x2019: "'", 0x201a: "'", 0x201b: "'", 0x02bc: "'", 0x201c: '"', 0x201d: '"', 0x201e: '"', 0x201f: '"'}
_LIGATURES = {0x0132: "IJ", 0x0133: "ij", 0x0149: "\u02bcn", 0x01f1: "DZ", 0x01f2: "Dz", 0x01f3: "dz", 0x01c4: "D\u017d",
              0x01c5: "D\u017e", 0x01c6: "d\u017e", 0x01c7: "LJ", 0x01c8: "Lj", 0x01c9: "lj", 0x01ca: "NJ", 0x01cb: "Nj",
              0x01cc: "nj", 0xfb00: "ff", 0xfb01: "fi", 0xfb02: "fl", 0xfb03: "ffi", 0xfb04: "ffl", 0xfb05: "\u017ft",
              0xfb06: "st"}
_WIDTH = {0x3000: " "}
_WIDTH.update({c: chr(c - 0xfee0) for c in range(0xff01, 0xff5f)})          # full-width ASCII block -> ASCII
_CONTROL = {c: None for c in list(range(0x00, 0x09)) + [0x0b] + list(range(0x0e, 0x20)) + [0x7f, 0xfeff] +
            list(range(0x206a, 0x2070)) + list(range(0xfff9, 0xfffd))}
_C1 = {c: bytes([c]).decode("cp1252", errors="ignore") or None for c in range(0x80, 0xa0)}
_TABLE = {**_CONTROL, **_WIDTH, **_LIGATURES, **_QUOTES}


def _fix_text_basic(text):
    """The deterministic subset of ftfy.fix_text's default configuration (see the module docstring)."""
    text = text.replace("\r\n", "\n").replace("\r", "\n").replace("\u2028", "\n").replace("\u2029", "\n").replace("\u0085", "\n")
    text = text.translate(_C1).translate(_TABLE)     # C1 controls first: \x93 -> a curly quote -> straight
    return unicodedata.normalize("NFC", text)


def basic_clean(text):
    """clip.simple_tokenizer.basic_clean: ftfy.fix_text, two html.unescape passes, strip."""
    text = _ftfy.fix_text(text) if _ftfy is not None else _fix_text_basic(text)
    return html.unescape(html.unescape(text)).strip()


@lru_cache()
def bytes_to_unicode():
    """Reversible byte -> printable unicode character table (the GPT-2 / CLIP byte alphabet)."""
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return dict(zip(bs, [chr(c) for c in cs]))


def _pairs(word):
    return {(a, b) for a, b in zip(word[:-1], word[1:])}


class SimpleTokenizer:
    def __init__(self, bpe_path, n_merges=49152 - 256 - 2):
        opener = gzip.open if str(bpe_path).endswith(".gz") else open
        with opener(bpe_path, "rt", encoding="utf-8") as f:
            lines = f.read().split("\n")
        merges = [tuple(m.split()) for m in lines[1:1 + n_merges] if len(m.split()) == 2]
        self.byte_encoder = bytes_to_unicode()
        vocab = list(self.byte_encoder.values())
        vocab = vocab + [v + "</w>" for v in vocab] + ["".join(m) for m in merges] + ["<|startoftext|>", "<|endoftext|>"]
        self.encoder = {t: i for i, t in enumerate(vocab)}
        self.decoder = {i: t for t, i in self.encoder.items()}
        self.byte_decoder = {v: k for k, v in self.byte_encoder.items()}
        self.ranks = {m: i for i, m in enumerate(merges)}
        self.cache = {"<|startoftext|>": "<|startoftext|>", "<|endoftext|>": "<|endoftext|>"}
        self.sot, self.eot = self.encoder["<|startoftext|>"], self.encoder["<|endoftext|>"]

    def bpe(self, token):
        if token in self.cache:
            return self.cache[token]
        word = tuple(token[:-1]) + (token[-1] + "</w>",)
        pairs = _pairs(word)
        if not pairs:
            return token + "</w>"
        while True:
            best = min(pairs, key=lambda p: self.ranks.get(p, float("inf")))
            if best not in self.ranks:
                break
            a, b = best
            out, i = [], 0
            while i < len(word):
                if i < len(word) - 1 and word[i] == a and word[i + 1] == b:
                    out.append(a + b)
                    i += 2
                else:
                    out.append(word[i])
                    i += 1
            word = tuple(out)
            if len(word) == 1:
                break
            pairs = _pairs(word)
        res = " ".join(word)
        self.cache[token] = res
        return res

    def encode(self, text):
        text = re.sub(r"\s+", " ", basic_clean(text)).strip().lower()
        ids = []
        for tok in _PAT.findall(text):
            tok = "".join(self.byte_encoder[b] for b in tok.encode("utf-8"))
            ids.extend(self.encoder[t] for t in self.bpe(tok).split(" "))
        return ids

    def decode(self, ids):
        """clip.simple_tokenizer.decode (main.py:913-916 writes the prompts of the logged batch with it)."""
        ids = [int(i) for i in ids]
        while ids and ids[-1] == 0:          # zero padding of tokenize()
            ids.pop()
        text = "".join(self.decoder[i] for i in ids)
        return bytearray(self.byte_decoder[c] for c in text).decode("utf-8", errors="replace").replace("</w>", " ")


_TOK = {}


def get_tokenizer(bpe_path=None):
    path = bpe_path or os.environ.get("FFVC_BPE_VOCAB")
    if not path or not os.path.exists(path):
        raise FileNotFoundError("CLIP BPE vocabulary not found: pass bpe_path or set FFVC_BPE_VOCAB to "
                                "bpe_simple_vocab_16e6.txt.gz (it cannot be downloaded offline); pre-tokenised .pkl "
                                "datasets and `synthetic:<n>` need no vocabulary")
    if path not in _TOK:
        _TOK[path] = SimpleTokenizer(path)
    return _TOK[path]


def tokenize(texts, context_length=77, truncate=False, bpe_path=None, tokenizer=None):
    """clip.tokenize: str | list[str] -> int64 (n, context_length)."""
    if isinstance(texts, str):
        texts = [texts]
    tk = tokenizer or get_tokenizer(bpe_path)
    out = torch.zeros(len(texts), context_length, dtype=torch.long)
    for i, t in enumerate(texts):
        ids = [tk.sot] + tk.encode(t) + [tk.eot]
        if len(ids) > context_length:
            if not truncate:
                raise RuntimeError(f"Input {t} is too long for context length {context_length}")
            ids = ids[:context_length]
            ids[-1] = tk.eot
        out[i, :len(ids)] = torch.tensor(ids)
    return out
