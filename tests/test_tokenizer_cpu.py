"""unimedvl_amd.tokenizer.Qwen2Tokenizer against golden vectors produced by the REFERENCE's own tokenizer class
(codes/modeling/qwen2/tokenization_qwen2.py) on a synthetic byte-level BPE vocabulary
(oracle/gen_tokenizer_golden.py -> tests/golden/tokenizer/): ids and decoded strings bit for bit, the special-token
registration of data_utils.add_special_tokens, plus round-trip properties on arbitrary unicode."""
import json
import os
import unicodedata

from hypothesis import given, settings, strategies as st

from conftest import GOLDEN
from unimedvl_amd.data_utils import add_special_tokens
from unimedvl_amd.tokenizer import Qwen2Tokenizer

DIR = os.path.join(GOLDEN, "tokenizer")


def load():
    tok = Qwen2Tokenizer.from_pretrained(DIR)
    return add_special_tokens(tok)


def test_special_token_registration_matches_reference():
    g = json.load(open(os.path.join(DIR, "cases.json"), encoding="utf-8"))
    tok, new_token_ids, num_new = load()
    assert num_new == g["num_new_tokens"] and new_token_ids == g["new_token_ids"]
    assert len(tok) == g["len"]
    assert tok.special_tokens_map == g["special_tokens_map"]


def test_encode_decode_match_reference():
    g = json.load(open(os.path.join(DIR, "cases.json"), encoding="utf-8"))
    tok, _, _ = load()
    assert len(g["cases"]) >= 30
    for c in g["cases"]:
        ids = tok.encode(c["text"])
        assert ids == c["ids"], (c["text"], ids, c["ids"])
        assert tok.decode(ids) == c["decoded"]
        import torch
        assert tok.decode(torch.tensor(ids, dtype=torch.int64)) == c["decoded"]


def test_answer_postprocessing_contract():
    """inferencer.py:277-278: decode(...).split('<|im_end|>')[0].split('<|im_start|>')[1]"""
    tok, nt, _ = load()
    ids = [nt["bos_token_id"]] + tok.encode("No pleural effusion.") + [nt["eos_token_id"]] + tok.encode("junk")
    assert tok.decode(ids).split("<|im_end|>")[0].split("<|im_start|>")[1] == "No pleural effusion."
    assert tok.convert_tokens_to_ids(["<|im_start|>", "<|vision_end|>"]) == [nt["bos_token_id"], nt["end_of_image"]]
    assert tok.decode(ids, skip_special_tokens=True).endswith("junk")


@settings(max_examples=200, deadline=None)
@given(st.text(max_size=60))
def test_round_trip_is_nfc(s):
    tok, _, _ = load()
    ids = tok.encode(s)
    assert all(isinstance(i, int) and 0 <= i < len(tok) for i in ids)
    # a lone surrogate cannot be encoded to UTF-8 at all; hypothesis' st.text() never produces one
    assert tok.decode(ids) == unicodedata.normalize("NFC", s)
