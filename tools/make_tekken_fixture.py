#!/usr/bin/env python3
"""tools/make_tekken_fixture.py — a small tekken.json in the layout of the REAL Mistral "Tekken" file.

The synthetic checkpoints carry a machine-friendly tekken.json (one line, ASCII strings that spell their own id).  The real file
(mistral-common's tekken.json, what voxtral_tokenizer.c:186-340 is written against) looks different, and every difference is a
way for a JSON scanner to go wrong: it is pretty-printed; "config" comes first and holds a regex full of backslashes and quotes;
every vocab entry has a "token_str" the loader must SKIP - with \\uXXXX escapes, surrogate pairs, escaped quotes and backslashes,
braces, brackets and commas inside strings, or null where the bytes are not valid UTF-8; token_bytes are arbitrary byte strings
(partial UTF-8 sequences, NUL, control bytes); special tokens carry "is_control" and come with \\u escapes and raw UTF-8 in
their strings; further objects with nested arrays follow the two arrays.

Writes tests/golden/tekken_real_layout.json (deterministic).  tests/test_host_cpu.py decodes every id with the host library and
with the reference's own tokenizer (oracle/_ref) and demands identical bytes.
usage: python tools/make_tekken_fixture.py
"""
import base64
import json
import os
import random

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    rng = random.Random(20260924)
    pieces = [b"\x00", b" ", b"\n", b"\t", b"\r\n", b"\"", b"\\", b"\\\"", b"{", b"}", b"[", b"]", b",", b":", b"},{", b"\"]}", b"/", b"\\u00e9",
              "é".encode(), " café".encode(), "naïve".encode(), "日本語".encode(), " 日".encode(), "€".encode(), "😀".encode(), " 👍🏽".encode(),
              b"\xe2\x82", b"\xf0\x9f", b"\x98\x80", b"\xc3", b"\xa9", b"\xff\xfe", b"\x7f", b"\x01\x02", b" the", b"ing", b" Hello", b"World",
              b"<s>", b"</s>", b"[INST]", b"null", b"true", b"\"rank\": 7", b"token_bytes", b" \xe2\x80\x94 ", "Ünïcödé".encode(), b"a" * 120]
    for b in range(256):
        pieces.append(bytes([b]))
    alphabet = [bytes([c]) for c in range(32, 127)] + ["é".encode(), "ß".encode(), "ж".encode(), "中".encode(), "🙂".encode()]
    while len(pieces) < 3000:
        n = rng.randint(1, 9)
        tok = b"".join(rng.choice(alphabet) for _ in range(n))
        if rng.random() < 0.15:
            tok = tok[:rng.randint(1, len(tok))]                    # may cut a multi-byte character in half
        pieces.append(tok)
    vocab = []
    for r, b in enumerate(pieces):
        try:
            ts = b.decode("utf-8")
        except UnicodeDecodeError:
            ts = None
        vocab.append({"rank": r, "token_bytes": base64.b64encode(b).decode(), "token_str": ts})
    special = []
    names = {0: "<unk>", 1: "<s>", 2: "</s>", 3: "[INST]", 4: "[/INST]", 5: "[AVAILABLE_TOOLS]", 6: "[/AVAILABLE_TOOLS]", 7: "[TOOL_RESULTS]",
             8: "[/TOOL_RESULTS]", 9: "[TOOL_CALLS]", 10: "[IMG]", 11: "<pad>", 12: "[IMG_BREAK]", 13: "[IMG_END]", 14: "[PREFIX]", 15: "[MIDDLE]",
             16: "[SUFFIX]", 17: "[SYSTEM_PROMPT]", 18: "[/SYSTEM_PROMPT]", 19: "[TOOL_CONTENT]", 24: "[AUDIO]", 25: "[BEGIN_AUDIO]",
             32: "[STREAMING_PAD]", 33: "[STREAMING_WORD]", 34: "[TRANSCRIBE]", 40: "<café \"q\" \\ x>", 41: "日本", 42: "tab\there"}
    for r in range(1000):
        special.append({"rank": r, "token_str": names.get(r, f"<SPECIAL_{r}>"), "is_control": r != 40})
    doc = {
        "config": {
            "pattern": "[^\\r\\n\\p{L}\\p{N}]?[\\p{Lu}\\p{Lt}\\p{Lm}\\p{Lo}\\p{M}]*[\\p{Ll}\\p{Lm}\\p{Lo}\\p{M}]+|\\p{N}| ?[^\\s\\p{L}\\p{N}]+[\\r\\n/]*|\\s*[\\r\\n]+|\\s+(?!\\S)|\\s+ \"{[,]}\"",
            "num_vocab_tokens": len(vocab), "default_vocab_size": 1000 + len(vocab), "default_num_special_tokens": 1000, "version": "v7",
        },
        "vocab": vocab,
        "special_tokens": special,
        "audio": {"sampling_rate": 16000, "frame_rate": 12.5, "encoding_config": {"num_mel_bins": 128, "hop_length": 160, "window_size": 400},
                  "chunk_length_s": None, "nested": [[1, 2, {"a": "]}\"", "b": [3, "\\"]}], "x,y"]},
        "multimodal": None,
    }
    path = os.path.join(ROOT, "tests", "golden", "tekken_real_layout.json")
    with open(path, "w", encoding="utf-8") as f:
        json.dump(doc, f, indent=2, ensure_ascii=True)       # \uXXXX escapes (surrogate pairs for the emoji) as json.dump of the real file gives
        f.write("\n")
    print(path, os.path.getsize(path), "bytes,", len(vocab), "vocab entries")


if __name__ == "__main__":
    main()
