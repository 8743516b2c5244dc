#!/usr/bin/env python3
"""Writes tests/golden/deflate_vectors.json: the known-answer vectors the reference's own tests hold
for the LZSS / Deflate / zlib / gzip encode path, as data (inputs + expected outputs).

Run in the build container (reads /root/reference/src to pull the long numeric tables out of the
tests so they are not transcribed by hand); the JSON it writes is what travels.
  lzss/encoder.rs:246-598      token vectors (the tests' comparison, window 0x10000, max 256)
  deflate/encoder.rs:660-1027  byte vectors / (value, bits) lists packed LSB first
  deflate/encoder.rs:1029-1210 (length, distance) -> (code, extra, extra bits) vectors
  zlib/encoder.rs:161-192, gzip/encoder.rs:147-164
"""
import json
import os
import re

REF = "/root/reference/src"
HERE = os.path.dirname(os.path.abspath(__file__))


def fn_body(path, name):
    src = open(os.path.join(REF, path)).read()
    i = src.index("fn %s()" % name)
    j = src.find("#[test]", i)
    return src[i:j if j > 0 else len(src)]


def bit_items(body):
    return [[int(v), int(n)] for v, n in re.findall(r"SmallBitVec::new\((\d+)(?:_u32)?,\s*(\d+)\)", body)]


def cyc(lo, hi, n):
    """input spec: n bytes cycling through lo..hi-1 (expanded by the test)"""
    return [["cycle", lo, hi, n]]


def lit(b):
    return [["bytes", list(b)]]


def rep(b, n):
    return [["repeat", b, n]]


def rl(tokens):
    """run-length form of a token list: [token, count]"""
    out = []
    for t in tokens:
        if out and out[-1][0] == t:
            out[-1][1] += 1
        else:
            out.append([t, 1])
    return out


S, R = "sym", "ref"
lzss = [
    dict(name="test_unit", input=lit(b"a"), tokens=[[S, 97]]),
    dict(name="test_2len", input=lit(b"aa"), tokens=[[S, 97], [S, 97]]),
    dict(name="test_3len", input=lit(b"aaa"), tokens=[[S, 97]] * 3),
    dict(name="test_4len", input=lit(b"aaaa"), tokens=[[S, 97], [R, 3, 0]]),
    dict(name="test_short_len", input=lit(b"a" * 11), tokens=[[S, 97], [R, 10, 0]]),
    dict(name="test_middle_repeat", input=lit(b"a" * 256), tokens=[[S, 97], [R, 255, 0]]),
    dict(name="test_long_repeat", input=lit(b"a" * 259), tokens=[[S, 97], [R, 256, 0], [S, 97], [S, 97]]),
    dict(name="test_long_repeat2", input=lit(b"a" * 260), tokens=[[S, 97], [R, 256, 0], [R, 3, 0]]),
    dict(name="test_5", input=lit(b"aaabbbaaabbb"), tokens=[[S, 97]] * 3 + [[S, 98]] * 3 + [[R, 6, 5]]),
    dict(name="test_6", input=lit(b"aabbaabbaaabbbaaabbbaabbaabb"),
         tokens=[[S, 97], [S, 97], [S, 98], [S, 98], [R, 6, 3], [S, 97], [S, 98], [R, 10, 5], [R, 6, 3]]),
    dict(name="test_7", input=cyc(0, 256, 0x10000), tokens=[[S, x] for x in range(256)] + [[R, 256, 255]] * 255),
    dict(name="test_8", input=cyc(0, 256, 768), tokens=[[S, x] for x in range(256)] + [[R, 256, 255]] * 2),
    dict(name="test_9", dict=cyc(0, 256, 256), input=cyc(0, 256, 512), tokens=[[R, 256, 255]] * 2),
    dict(name="test_10", dict=cyc(0, 256, 0x10001), input=cyc(0, 256, 512), tokens=[[R, 256, 256], [R, 256, 255]]),
    dict(name="test_11", input=lit(b"abc") + rep(100, 0x10000 - 3) + lit(b"abc"),
         tokens=[[S, 97], [S, 98], [S, 99], [S, 100]] + [[R, 256, 0]] * 255 + [[R, 252, 0], [R, 3, 65535]]),
]
for v in lzss:
    v["tokens"] = rl(v["tokens"])
    v.update(window=0x10000, max_match=256, min_match=3, lazy=3, comparison="lzss_tests")

enc = "deflate/encoder.rs"
arr3 = [[1, 1], [0, 2], [0, 5], [112, 16], [65423, 16]] + [[x, 8] for x in range(144, 256)]
deflate = [
    dict(name="test_empty", input=lit(b""), bytes=[3, 0]),
    dict(name="test_unit", input=lit(b"a"), bytes=[0x4B, 0x04, 0]),
    dict(name="test_arr", input=lit(b"a" * 11), bytes=[0x4B, 0x44, 0, 0]),
    dict(name="test_arr2", input=lit(b"aabbaabbaaabbbaaabbbaabbaabb"), bits=bit_items(fn_body(enc, "test_arr2"))),
    dict(name="test_arr3", input=cyc(144, 256, 112), bits=arr3),
    dict(name="test_arr4", input=cyc(144, 256, 224), bits=bit_items(fn_body(enc, "test_arr4"))),
]

# test_defaltelzsscode: (len, pos) -> (len code, len extra, bits, dist code, dist extra, bits)
body = fn_body(enc, "test_defaltelzsscode")
codes = []
for m in re.finditer(r"Reference \{ len: (\d+), pos: (\d+) \},.*?len: (\d+),\s*len_sub: SmallBitVec::new\((\d+), (\d+)\),\s*"
                     r"pos: (\d+),\s*pos_sub: SmallBitVec::new\((\d+), (\d+)\)", body, re.S):
    codes.append([int(x) for x in m.groups()])

containers = [
    dict(name="zlib_test_unit", kind="zlib", input=lit(b"a"), bytes=[0x78, 0xDA, 0x4B, 0x04, 0x00, 0x00, 0x62, 0x00, 0x62]),
    dict(name="zlib_test_unit_with_dict", kind="zlib", dict=lit(b"a"), input=lit(b"a"),
         bytes=[0x78, 0xF9, 0x00, 0x62, 0x00, 0x62, 0x4B, 0x04, 0x00, 0x00, 0x62, 0x00, 0x62]),
    dict(name="gzip_test_unit", kind="gzip", input=lit(b"a"),
         bytes=[0x1f, 0x8b, 0x08, 0, 0, 0, 0, 0, 0, 0xFF, 0x4b, 0x04, 0x00, 0x43, 0xbe, 0xb7, 0xe8, 0x01, 0, 0, 0]),
]
checksums = dict(crc32_ieee_reverse=dict(input=lit(b"123456789"), value=0xCBF43926))

out = dict(lzss=lzss, deflate=deflate, codes=codes, containers=containers, checksums=checksums)
with open(os.path.join(HERE, "deflate_vectors.json"), "w") as f:
    json.dump(out, f, separators=(",", ":"))
print("lzss", len(lzss), "deflate", len(deflate), "codes", len(codes), "containers", len(containers),
      "arr2 items", len(deflate[3]["bits"]), "arr4 items", len(deflate[5]["bits"]))
