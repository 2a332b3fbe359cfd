"""Toy tokenizer for the parity tests (TEST INFRASTRUCTURE).  The real Qwen2 BPE vocabulary
ships with the checkpoint, which is not available offline.  Text is a space-separated list
of integers (other words hash to ids); special ids render as the markers the reference's post-processing splits on
(codes/inferencer.py:277-278)."""


class ToyTokenizer:
    def __init__(self, new_token_ids):
        self.names = {new_token_ids["bos_token_id"]: "<|im_start|>", new_token_ids["eos_token_id"]: "<|im_end|>",
                      new_token_ids["start_of_image"]: "<|vision_start|>", new_token_ids["end_of_image"]: "<|vision_end|>"}

    def encode(self, s):
        """integers stand for themselves; any other word (the reference's English think prompts, inferencer.py:23-28)
        hashes to a stable id in [5, 290)"""
        import zlib
        return [int(x) if x.lstrip("-").isdigit() else zlib.crc32(x.encode()) % 285 + 5 for x in s.split()]

    def decode(self, ids):
        out = []
        for i in ids:
            i = int(i)
            out.append(self.names.get(i, f" {i}"))
        return "".join(out)
