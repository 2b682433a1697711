"""Debug aid: byte-aligned probing vs the oracle on one generated text (GPU)."""
import os, random, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from test_gpu_parity import gpu_scan_buffer, oracle_hits

pats = ["ERROR", "WARN", "foo", "(?i)Fail", "a\\.b", "panic: [a-z]+", "x=\\d+;", "status=5[0-9]{2}", "\\bGET\\b /api"]
flags = [14, 14, 10, 14, 14, 6, 14, 14, 14]
ids = [0, 1, 2, 3, 4, 5, 6, 7, 7]
rng = random.Random(78)
words = [b"ERROR", b"WARN", b"foo", b"FAIL", b"fAiL", b"a.b", b"panic: oops", b"x=12;", b"ERRO", b"WAR", b"fo", b"fai", b"x=;", b"foofoo",
         b"status=503", b"status=200", b"GET /api", b"GETS /api"]
out = bytearray()
while len(out) < 300000:
    line = bytearray()
    for _ in range(rng.choice([0, 2, 6, 6, 40])):
        line += rng.choice(words) if rng.random() < 0.4 else bytes(rng.choice(b"abcdefoOrRE =.;0123") for _ in range(rng.randint(1, 9)))
        if rng.random() < 0.5:
            line += b" "
    if rng.random() < 0.02:
        line[len(line) // 2:len(line) // 2] = b"\0"
    out += line + b"\n"
out[100000:100000] = b"z" * 30000 + b" ERROR foo " + b"y" * 20000
for at in (1021, 1022, 1023, 2045, 16381, 16382, 16383, 32765, 65533):
    out[at:at + 5] = b"ERROR"
    out[at + 3000:at + 3003] = b"foo"
data = bytes(out[:290000]) + rng.choice([b"foo", b"WARN", b"ERROR\n", b"fo"])
for bs in (262140, 4096):
    want, nlines = oracle_hits(data, pats, flags, ids, buffer_size=bs)
    got, stats = gpu_scan_buffer(torch, data, pats, flags, ids, buffer_size=bs)
    print(bs, os.environ.get("HG_NO_BYTE_WINDOWS"), len(got), len(want), got == want, stats)
    if got != want:
        sg, sw = set(got), set(want)
        print(" missing", sorted(sw - sg)[:8], len(sw - sg))
        print(" extra", sorted(sg - sw)[:8], len(sg - sw))
        for h in sorted(sg - sw)[:3]:
            print("  extra line:", data[h[3]:h[3] + h[4]][:200])
