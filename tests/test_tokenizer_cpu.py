"""CLIP BPE tokenizer front-end (tokenizer.py) on a tiny hand-made merges file: merge order, </w> handling, SOT/EOT
framing, zero padding, truncate=True forcing EOT at the last position, int64 output (main.py:733 dispatches on it)."""
import gzip

import pytest
import torch

from feed_forward_vqgan_clip_amd import tokenizer as T


@pytest.fixture()
def tk(tmp_path):
    p = tmp_path / "bpe.txt.gz"
    merges = ["#version: test", "c a", "ca t</w>", "d o", "do g</w>", "a t</w>"]
    with gzip.open(p, "wt", encoding="utf-8") as f:
        f.write("\n".join(merges) + "\n")
    return T.SimpleTokenizer(str(p))


def test_vocab_layout(tk):
    # 256 byte symbols, 256 word-final byte symbols, the merges, then SOT / EOT
    assert len(tk.encoder) == 512 + 5 + 2
    assert tk.sot == 517 and tk.eot == 518
    assert tk.encoder["cat</w>"] == 513 and tk.encoder["dog</w>"] == 515


def test_bpe_merges_by_rank(tk):
    assert tk.bpe("cat") == "cat</w>"
    assert tk.bpe("dog") == "dog</w>"
    assert tk.bpe("at") == "at</w>"
    assert tk.bpe("bat") == "b at</w>"              # 'a t</w>' applies, nothing merges 'b'
    assert tk.bpe("cab") == "ca b</w>"


def test_encode_decode_roundtrip(tk):
    ids = tk.encode("  A  Cat &amp; a DOG's  ")
    assert tk.decode(ids).strip() == "a cat & a dog 's"
    assert tk.encoder["cat</w>"] in ids and tk.encoder["dog</w>"] in ids


def test_tokenize_framing_padding_truncation(tk):
    out = T.tokenize(["cat", "dog cat dog cat dog"], context_length=6, truncate=True, tokenizer=tk)
    assert out.dtype == torch.long and tuple(out.shape) == (2, 6)
    assert out[0].tolist() == [tk.sot, 513, tk.eot, 0, 0, 0]
    assert out[1, 0] == tk.sot and out[1, -1] == tk.eot and (out[1] != 0).all()
    with pytest.raises(RuntimeError):
        T.tokenize(["dog cat dog cat dog"], context_length=6, truncate=False, tokenizer=tk)


def test_missing_vocabulary_is_loud(monkeypatch):
    monkeypatch.delenv("FFVC_BPE_VOCAB", raising=False)
    with pytest.raises(FileNotFoundError):
        T.tokenize(["a cat"])


def test_load_dataset_tokenises_text_files(tk, tmp_path, monkeypatch):
    from feed_forward_vqgan_clip_amd import main as fmain
    p = tmp_path / "prompts.txt"
    p.write_text("cat\ndog\n")
    vocab = tmp_path / "bpe.txt.gz"
    toks = fmain.load_dataset(str(p), bpe_path=str(vocab))
    assert tuple(toks.shape) == (2, 77) and toks.dtype == torch.long and toks[1, 1] == 515
    out = tmp_path / "tok.pkl"
    fmain.tokenize(str(p), out=str(out), bpe_path=str(vocab))
    assert torch.equal(fmain.load_dataset(str(out)), toks)


def test_unicode_letters_and_numbers_follow_clip_pattern(tk):
    r"""clip.simple_tokenizer's pre-token pattern uses \p{L} / \p{N}: an accented or non-Latin word is ONE pre-token (its
    UTF-8 bytes then go through the byte alphabet and BPE), every unicode digit is its own pre-token."""
    text = "café naïve 東京2024 привет's ½ x²"
    assert T._PAT.findall(text) == ["café", "naïve", "東京", "2", "0", "2", "4",
                                    "привет", "'s", "½", "x", "²"]
    be = tk.byte_encoder
    e_acute = [be[b] for b in "é".encode("utf-8")]                    # two byte symbols
    ids = tk.encode("Café")
    # one word: 'c a' merges (rank 0), then f and the two bytes of e-acute, the LAST carrying the end-of-word marker
    assert ids == [tk.encoder["ca"], tk.encoder["f"], tk.encoder[e_acute[0]], tk.encoder[e_acute[1] + "</w>"]]
    # the ASCII-only pattern of round 2 split it into 'caf' + the accent (f would have carried the marker)
    assert tk.encoder["f</w>"] not in ids
    assert tk.decode(ids).strip() == "café"
    # CJK and Cyrillic round-trip through the byte alphabet
    for t in ("東京", "привет мир", "x² + ½"):
        assert tk.decode(tk.encode(t)).replace(" ", "") == t.replace(" ", "")


def test_basic_clean_subset_of_ftfy():
    """Without ftfy the deterministic part of its default configuration is applied (documented gap: mojibake repair)."""
    if T._ftfy is not None:
        pytest.skip("ftfy present: upstream's own cleaner is used")
    assert T.basic_clean("  “Quoted” ‘x’ ﬁne ＡＢＣ　ok\x07 &amp;amp; ") == "\"Quoted\" 'x' fine ABC ok &"
    assert T.basic_clean("a\r\nb c") == "a\nb\nc"
    assert T.basic_clean("é") == "é"                       # NFC
    assert T.basic_clean("\x93x\x94") == "\"x\""                      # C1 controls -> cp1252 curly quotes -> straight


def test_ids_equal_hf_clip_tokenizer_on_a_synthetic_vocabulary(tmp_path):
    """Second witness for the BPE front-end (the real vocabulary is absent offline, so the ids cannot be pinned to
    clip.tokenize itself): HuggingFace's CLIPTokenizer — an independent port of the same OpenAI algorithm (byte alphabet, </w> marker,
    merges by rank, the pre-token pattern, SOT / EOT framing) — given a synthetic merges file laid out the way CLIP's vocabulary is
    (256 byte symbols, 256 word-final ones, the merges, the two specials) must produce the same ids for the same text."""
    transformers = pytest.importorskip("transformers")
    import json
    import random
    rnd = random.Random(7)
    words = ["cat", "dog", "photo", "painting", "of", "a", "the", "red", "blue", "castle", "river", "night", "2024", "don't", "it's",
             "sun", "sunset", "mountain", "mountains", "artstation", "trending", "oil", "on", "canvas", "hd", "4k", "ocean", "waves"]
    # a merges table that actually fires on these words: walk each word, merge adjacent symbols left to right, in a shuffled order
    merges, seen, joined = [], set(), set()
    for w in words * 2:
        sym = list(w[:-1]) + [w[-1] + "</w>"]
        while len(sym) > 1:
            i = rnd.randrange(len(sym) - 1)
            pair = (sym[i], sym[i + 1])
            if pair not in seen and "'" not in pair[0] + pair[1] and pair[0] + pair[1] not in joined:   # (joined strings unique, as in CLIP's file)
                seen.add(pair)
                joined.add(pair[0] + pair[1])
                merges.append(pair)
            sym[i:i + 2] = [sym[i] + sym[i + 1]]
    mpath = tmp_path / "merges.txt"
    mpath.write_text("#version: 0.2\n" + "\n".join(f"{a} {b}" for a, b in merges) + "\n", encoding="utf-8")
    mine = T.SimpleTokenizer(str(mpath))
    vocab = {tok: i for tok, i in mine.encoder.items()}                  # same construction rule -> same ids by definition of the format
    assert vocab["<|startoftext|>"] == len(vocab) - 2 and vocab["<|endoftext|>"] == len(vocab) - 1
    vpath = tmp_path / "vocab.json"
    vpath.write_text(json.dumps(vocab), encoding="utf-8")
    hf = transformers.CLIPTokenizer(str(vpath), str(mpath))
    texts = ["a photo of a cat", "the red castle on the river at night, trending on artstation", "it's a dog, don't panic!",
             "oil painting of mountains   at  sunset 2024 hd 4k", "ocean waves", "a", "sunsets over the mountain... (oil on canvas)",
             "catdog dogcat photophoto"]
    for t in texts:
        ref = hf(t)["input_ids"]
        ids = [mine.sot] + mine.encode(t) + [mine.eot]
        assert ids == ref, (t, ids, ref)
    out = T.tokenize(texts, context_length=32, truncate=True, tokenizer=mine)
    ref = hf(texts, padding="max_length", max_length=32, truncation=True)["input_ids"]
    for row, r in zip(out.tolist(), ref):
        n = r.index(mine.eot) + 1
        assert row[:n] == r[:n] and all(v == 0 for v in row[n:])        # clip.tokenize pads with 0, HF with its pad id
