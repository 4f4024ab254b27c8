#!/usr/bin/env python3
"""Records tests/golden/huffman_limit.json: the digest of jpeg_amd_huffman_build's tables for the seeded, skewed histograms
of tests/test_huffman_limit_cpu.py.  Run against a build whose tables are trusted -- it was recorded with round 5's library
(JPEG_AMD_LIBRARY=<round-5 libjpeg_amd.so>), the one whose files equal the reference's byte for byte."""
import hashlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_huffman_limit_cpu import build, histograms
SEED, COUNT = 20240807, 3000
h = hashlib.sha256(); limited = 0
for f in histograms(SEED, COUNT):
    counts, values = build(f)
    limited += counts[15] > 0
    h.update(bytes(counts)); h.update(bytes(values))
out = {"seed": SEED, "count": COUNT, "tables_with_16_bit_codes": limited, "sha256": h.hexdigest(),
       "recorded_with": os.environ.get("JPEG_AMD_LIBRARY", "the product build")}
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "huffman_limit.json"), "w"), indent=1)
print(out)
