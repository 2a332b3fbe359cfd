"""Qwen2 byte-level BPE tokenizer (host side), standing in for the reference's
``codes/modeling/qwen2/tokenization_qwen2.py`` (a copy of the Hugging Face *slow* Qwen2Tokenizer) so that the engine
needs nothing from `transformers` at run time.  Same files, same ids:

  * `vocab.json` (token string -> id) and `merges.txt` (one merge per line, rank = line order) from the checkpoint
    directory, plus the added / special tokens of `tokenizer_config.json` (`added_tokens_decoder`) or `added_tokens.json`;
  * text -> NFC (tokenization_qwen2.py:321-323) -> split on added tokens, leftmost-longest, never split further ->
    the pre-tokenisation regex (tokenization_qwen2.py:37) -> UTF-8 bytes mapped to the 256 printable stand-in characters
    (tokenization_qwen2.py:40-62) -> byte-pair merges, lowest rank first, every occurrence left to right
    (tokenization_qwen2.py:211-251) -> ids;
  * decode: added tokens verbatim, everything else through the inverse byte map and UTF-8 with errors="replace"
    (tokenization_qwen2.py:276-296); no clean-up of spaces, no spaces between special tokens.

Only the surface the reference's call sites use is provided (`encode`, `decode`, `add_tokens`, `convert_tokens_to_ids`,
`special_tokens_map`, `len`; data_utils.py:140-175, bagel.py:391, inferencer.py:277-278).  Parity is pinned against the
reference's own class on a synthetic vocabulary: oracle/gen_tokenizer_golden.py -> tests/golden/tokenizer/.
"""
import json
import os
import unicodedata
from typing import Dict, Iterable, List, Optional, Sequence, Tuple, Union

import regex

# the Qwen2 pre-tokenisation pattern (tokenization_qwen2.py:37): contractions, letter runs with one optional leading
# non-letter, SINGLE digits, punctuation runs with an optional leading space, newline runs, trailing / other whitespace
_PRETOKENIZE = regex.compile(
    r"(?i:'s|'t|'re|'ve|'m|'ll|'d)|[^\r\n\p{L}\p{N}]?\p{L}+|\p{N}| ?[^\s\p{L}\p{N}]+[\r\n]*|\s*[\r\n]+|\s+(?!\S)|\s+")


def _byte_alphabet() -> Dict[int, str]:
    """byte value -> stand-in character: printable Latin-1 bytes map to themselves, the other 68 to U+0100.. in order"""
    keep = [b for b in range(256) if 33 <= b <= 126 or 161 <= b <= 172 or 174 <= b <= 255]
    table, spill = {}, 0
    for b in range(256):
        if b in keep:
            table[b] = chr(b)
        else:
            table[b] = chr(256 + spill)
            spill += 1
    return table


_BYTE_TO_CHAR = _byte_alphabet()
_CHAR_TO_BYTE = {c: b for b, c in _BYTE_TO_CHAR.items()}


class Qwen2Tokenizer:
    def __init__(self, vocab: Dict[str, int], merges: Sequence[Tuple[str, str]], added_tokens: Optional[Dict[str, int]] = None,
                 eos_token: str = "<|endoftext|>", pad_token: str = "<|endoftext|>", unk_token: str = "<|endoftext|>",
                 bos_token: Optional[str] = None, additional_special_tokens: Iterable[str] = ()):
        self.encoder = dict(vocab)
        self.decoder = {i: t for t, i in self.encoder.items()}
        self.ranks = {tuple(m): r for r, m in enumerate(merges)}
        self.added: Dict[str, int] = {}
        self.added_by_id: Dict[int, str] = {}
        self._special_names = {"bos_token": bos_token, "eos_token": eos_token, "unk_token": unk_token, "pad_token": pad_token}
        self._additional = list(additional_special_tokens)
        self._cache: Dict[str, Tuple[str, ...]] = {}
        self._by_first = None
        for tok, idx in (added_tokens or {}).items():
            self._register(tok, idx)
        # named special tokens that are neither in the vocabulary nor added yet get fresh ids, like PreTrainedTokenizer
        for tok in [t for t in self._special_names.values() if t] + self._additional:
            if tok not in self.added and tok not in self.encoder:
                self._register(tok, len(self))

    # ------------------------------------------------------------------ construction
    @classmethod
    def from_files(cls, vocab_file: str, merges_file: str, tokenizer_config: Optional[str] = None,
                   added_tokens_file: Optional[str] = None) -> "Qwen2Tokenizer":
        with open(vocab_file, encoding="utf-8") as f:
            vocab = json.load(f)
        merges = []
        with open(merges_file, encoding="utf-8") as f:
            for i, line in enumerate(f):
                line = line.strip()
                if not line or (i == 0 and line.startswith("#version:")):
                    continue
                a, b = line.split()
                merges.append((a, b))
        added, kw = {}, {}
        if added_tokens_file and os.path.exists(added_tokens_file):
            with open(added_tokens_file, encoding="utf-8") as f:
                added.update({t: int(i) for t, i in json.load(f).items()})
        if tokenizer_config and os.path.exists(tokenizer_config):
            with open(tokenizer_config, encoding="utf-8") as f:
                cfg = json.load(f)
            for idx, spec in (cfg.get("added_tokens_decoder") or {}).items():
                added[spec["content"] if isinstance(spec, dict) else str(spec)] = int(idx)
            for name in ("bos_token", "eos_token", "unk_token", "pad_token"):
                if name in cfg:
                    v = cfg[name]
                    kw[name] = v["content"] if isinstance(v, dict) else v
            extra = cfg.get("additional_special_tokens") or []
            kw["additional_special_tokens"] = [e["content"] if isinstance(e, dict) else e for e in extra]
        return cls(vocab, merges, added, **kw)

    @classmethod
    def from_pretrained(cls, path: str) -> "Qwen2Tokenizer":
        """the checkpoint directory the reference passes to Qwen2Tokenizer.from_pretrained (interactive_vqa_inferencer.py:236)"""
        j = lambda n: os.path.join(path, n)   # noqa: E731
        return cls.from_files(j("vocab.json"), j("merges.txt"), j("tokenizer_config.json"), j("added_tokens.json"))

    def _register(self, tok: str, idx: int):
        self.added[tok] = idx
        self.added_by_id[idx] = tok
        self._by_first = None

    # ------------------------------------------------------------------ the surface the reference uses
    def __len__(self):
        ids = set(self.encoder.values()) | set(self.added_by_id)
        return max(ids) + 1 if ids else 0

    @property
    def vocab_size(self):
        return len(self.encoder)

    @property
    def special_tokens_map(self) -> Dict[str, Union[str, List[str]]]:
        out = {k: v for k, v in self._special_names.items() if v}
        if self._additional:
            out["additional_special_tokens"] = list(self._additional)
        return out

    def add_tokens(self, new_tokens: Union[str, Sequence[str]], special_tokens: bool = False) -> int:
        if isinstance(new_tokens, str):
            new_tokens = [new_tokens]
        n = 0
        for tok in new_tokens:
            if tok in self.added or (tok in self.encoder and not special_tokens):
                continue
            self._register(tok, self.encoder[tok] if tok in self.encoder else len(self))
            n += 1
        return n

    def convert_tokens_to_ids(self, tokens: Union[str, Sequence[str]]):
        if isinstance(tokens, str):
            return self._token_id(tokens)
        return [self._token_id(t) for t in tokens]

    def convert_ids_to_tokens(self, ids: Union[int, Sequence[int]]):
        if isinstance(ids, int):
            return self.added_by_id.get(ids, self.decoder.get(ids))
        return [self.added_by_id.get(int(i), self.decoder.get(int(i))) for i in ids]

    def _token_id(self, tok: str):
        if tok in self.added:
            return self.added[tok]
        if tok in self.encoder:
            return self.encoder[tok]
        unk = self._special_names.get("unk_token")
        return self.added.get(unk, self.encoder.get(unk))

    # ------------------------------------------------------------------ encode
    def _split_on_added(self, text: str) -> List[Tuple[bool, str]]:
        """[(is_added, piece)]: leftmost match wins, the longest added token at that position"""
        if not self.added:
            return [(False, text)] if text else []
        if self._by_first is None:
            by_first: Dict[str, List[str]] = {}
            for tok in self.added:
                if tok:
                    by_first.setdefault(tok[0], []).append(tok)
            for lst in by_first.values():
                lst.sort(key=len, reverse=True)
            self._by_first = by_first
        out, start, i, n = [], 0, 0, len(text)
        while i < n:
            hit = None
            for tok in self._by_first.get(text[i], ()):
                if text.startswith(tok, i):
                    hit = tok
                    break
            if hit is None:
                i += 1
                continue
            if i > start:
                out.append((False, text[start:i]))
            out.append((True, hit))
            i += len(hit)
            start = i
        if start < n:
            out.append((False, text[start:]))
        return out

    def _bpe(self, word: str) -> Tuple[str, ...]:
        got = self._cache.get(word)
        if got is not None:
            return got
        parts = list(word)
        while len(parts) > 1:
            best_rank, best = None, None
            for pair in zip(parts, parts[1:]):
                rk = self.ranks.get(pair)
                if rk is not None and (best_rank is None or rk < best_rank):
                    best_rank, best = rk, pair
            if best is None:
                break
            a, b = best
            merged, i = [], 0
            while i < len(parts):
                if i + 1 < len(parts) and parts[i] == a and parts[i + 1] == b:
                    merged.append(a + b)
                    i += 2
                else:
                    merged.append(parts[i])
                    i += 1
            parts = merged
        out = tuple(parts)
        self._cache[word] = out
        return out

    def tokenize(self, text: str) -> List[str]:
        tokens: List[str] = []
        # the reference normalises the whole input first (prepare_for_tokenization), then splits on added tokens
        for is_added, piece in self._split_on_added(unicodedata.normalize("NFC", text)):
            if is_added:
                tokens.append(piece)
                continue
            for chunk in _PRETOKENIZE.findall(piece):
                tokens.extend(self._bpe("".join(_BYTE_TO_CHAR[b] for b in chunk.encode("utf-8"))))
        return tokens

    def encode(self, text: str, add_special_tokens: bool = True) -> List[int]:
        """Qwen2 has no BOS / EOS template: add_special_tokens changes nothing, as in the reference."""
        return [self._token_id(t) for t in self.tokenize(text)]

    # ------------------------------------------------------------------ decode
    def decode(self, token_ids, skip_special_tokens: bool = False, **_unused) -> str:
        if hasattr(token_ids, "tolist"):
            token_ids = token_ids.tolist()
        if isinstance(token_ids, int):
            token_ids = [token_ids]
        specials = set(t for t in self._special_names.values() if t) | set(self._additional)
        pieces, run = [], []

        def flush():
            if run:
                pieces.append(bytes(_CHAR_TO_BYTE[c] for c in "".join(run)).decode("utf-8", errors="replace"))
                run.clear()
        for i in token_ids:
            i = int(i)
            if i in self.added_by_id:
                tok = self.added_by_id[i]
                if skip_special_tokens and tok in specials:
                    continue
                flush()
                pieces.append(tok)
            else:
                tok = self.decoder.get(i)
                if tok is not None:
                    run.append(tok)
        flush()
        return "".join(pieces)
