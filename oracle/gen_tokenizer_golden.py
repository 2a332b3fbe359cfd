"""TEST INFRASTRUCTURE ONLY.  Golden vectors for unimedvl_amd/tokenizer.py, produced by the REFERENCE's own tokenizer
class (codes/modeling/qwen2/tokenization_qwen2.py, loaded from /root/reference in this container only).

The real Qwen2 vocabulary is not available offline, so the fixture vocabulary is synthetic: a byte-level BPE trained
right here on a small embedded corpus (256 byte tokens + N merges).  The reference class is instantiated on those
files, the UniMedVL special tokens are added the way data_utils.add_special_tokens does (data_utils.py:140-175), and its
encode / decode results on a list of probe strings are stored next to the vocabulary:

    tests/golden/tokenizer/vocab.json, merges.txt, tokenizer_config.json      (synthetic data made by this script)
    tests/golden/tokenizer/cases.json                                          ([{text, ids, decoded}], new_token_ids)

Run:  PYTHONDONTWRITEBYTECODE=1 python -m oracle.gen_tokenizer_golden
"""
import collections
import importlib.util
import json
import os

import regex

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "tokenizer")
REF = "/root/reference/codes/modeling/qwen2/tokenization_qwen2.py"

CORPUS = """The chest X-ray shows mild cardiomegaly with clear lung fields. What abnormality is visible in this image?
Is there evidence of pleural effusion? No pleural effusion or pneumothorax is seen. The heart size is at the upper limit of normal.
A fundus photograph of the left eye showing diabetic retinopathy; it's moderate, isn't it? We'll compare with 2019, 2021 and 2024.
Generate a histopathology image of colon tissue, H&E stain, 20x magnification.  Answer: adenocarcinoma (grade 2/3).
影像显示心脏扩大，双肺纹理清晰。请描述这张图像中的异常。 胸部X光片未见明显异常。
Röntgen-Thorax: keine Auffälligkeiten — naïve café déjà vu. Ελληνικά, русский текст, العربية, 日本語のテキスト。
def f(x):\n\treturn x**2 + 1  # comment\n\n\nTabs\tand   multiple   spaces   \n trailing space \r\n windows line
é composed vs é; emoji 🩻🫀 and symbols ±≤≥µm ½ ① ²
"""

PROBES = [
    "", " ", "  ", "\n", "a", "The chest X-ray shows cardiomegaly.", "What abnormality is visible?", "it's isn't we'll I'M THEY'RE you'd",
    "12345 67 8", "2019-2024: 3.5cm x 2.1cm", "hello   world  \n\n  next", "trailing space ", " leading", "\ttab\tseparated\r\nlines\n",
    "影像显示心脏扩大。", "日本語のテキスト", "русский текст и English mixed", "naïve café déjà vu", "é vs é", "🩻 emoji 🫀!",
    "<|im_start|>user\nDescribe the image.<|im_end|>\n<|im_start|>assistant\n", "<|vision_start|><|vision_end|>",
    "text<|im_end|>more<|endoftext|>", "<|im_start|><|im_start|>", "<|im_st", "a<|vision_start|>b <|vision_end|> c",
    "You should first think about the planning process in the mind and then generate the image.",
    "±≤≥µm ½ ① ² — … “quotes”", "x" * 70, "ab" * 40 + " " * 9 + "c", "A B C　D",
]


def train_bpe(corpus, n_merges, byte_to_char, pattern):
    words = collections.Counter()
    for chunk in pattern.findall(corpus):
        words[tuple(byte_to_char[b] for b in chunk.encode("utf-8"))] += 1
    vocab = {c: i for i, c in enumerate(sorted(set(byte_to_char.values())))}
    merges = []
    for _ in range(n_merges):
        pairs = collections.Counter()
        for w, c in words.items():
            for a, b in zip(w, w[1:]):
                pairs[(a, b)] += c
        if not pairs:
            break
        (a, b), _ = max(pairs.items(), key=lambda kv: (kv[1], kv[0]))
        if a + b in vocab:
            break
        merges.append((a, b))
        vocab[a + b] = len(vocab)
        new_words = collections.Counter()
        for w, c in words.items():
            out, i = [], 0
            while i < len(w):
                if i + 1 < len(w) and w[i] == a and w[i + 1] == b:
                    out.append(a + b)
                    i += 2
                else:
                    out.append(w[i])
                    i += 1
            new_words[tuple(out)] += c
        words = new_words
    return vocab, merges


def main():
    spec = importlib.util.spec_from_file_location("ref_tokenization_qwen2", REF)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    os.makedirs(OUT, exist_ok=True)
    vocab, merges = train_bpe(CORPUS, 400, ref.bytes_to_unicode(), regex.compile(ref.PRETOKENIZE_REGEX))
    with open(os.path.join(OUT, "vocab.json"), "w", encoding="utf-8") as f:
        json.dump(vocab, f, ensure_ascii=False)
    with open(os.path.join(OUT, "merges.txt"), "w", encoding="utf-8") as f:
        f.write("#version: 0.2\n" + "".join(f"{a} {b}\n" for a, b in merges))
    tok = ref.Qwen2Tokenizer(os.path.join(OUT, "vocab.json"), os.path.join(OUT, "merges.txt"))
    base_added = dict(tok.added_tokens_encoder)   # <|endoftext|> got the first free id
    with open(os.path.join(OUT, "tokenizer_config.json"), "w", encoding="utf-8") as f:
        json.dump({"added_tokens_decoder": {str(i): {"content": t, "special": True} for t, i in base_added.items()},
                   "eos_token": "<|endoftext|>", "pad_token": "<|endoftext|>", "unk_token": "<|endoftext|>"}, f)
    # data_utils.add_special_tokens (data_utils.py:140-175)
    present = []
    for v in tok.special_tokens_map.values():
        present += [v] if isinstance(v, str) else list(v)
    wanted = ["<|im_start|>", "<|im_end|>", "<|vision_start|>", "<|vision_end|>"]
    num_new = tok.add_tokens([t for t in wanted if t not in present])
    new_token_ids = dict(zip(["bos_token_id", "eos_token_id", "start_of_image", "end_of_image"],
                             [tok.convert_tokens_to_ids(t) for t in wanted]))
    cases = []
    for text in PROBES:
        ids = tok.encode(text)
        cases.append({"text": text, "ids": ids, "decoded": tok.decode(ids)})
    with open(os.path.join(OUT, "cases.json"), "w", encoding="utf-8") as f:
        json.dump({"num_new_tokens": num_new, "new_token_ids": new_token_ids, "len": len(tok),
                   "special_tokens_map": tok.special_tokens_map, "cases": cases}, f, ensure_ascii=False, indent=0)
    print(f"vocab {len(vocab)} merges {len(merges)} added {base_added} new {new_token_ids}; {len(cases)} cases -> {OUT}")


if __name__ == "__main__":
    main()
