#!/usr/bin/env python3
"""Generates kjarni_amd/csrc/unicode_tables.inc.

The reference tokenises with the HF `tokenizers` crate 0.22.1 (Cargo.toml:34,
call sites crates/kjarni-transformers/src/cpu/encoder/traits.rs:141-145).  Its
BertNormalizer / BertPreTokenizer depend on Unicode property tables compiled
into that crate (unicode_categories, unicode-normalization-alignments, Rust's
char::is_whitespace / to_lowercase).  To pin token ids bit-exactly, the tables
are not taken from some other Unicode database: they are PROBED, code point by
code point, out of the same Rust core through its Python binding
(`tokenizers` 0.22.2, the only build of the crate available offline):

  clean_text      BertNormalizer(clean_text only)       -> keep / drop / -> ' '
  NFD             normalizers.NFD()                      -> canonical decomposition
  Mn              BertNormalizer(strip_accents only)     -> dropped after NFD
  lowercase       normalizers.Lowercase()                -> per-char to_lowercase
  split class     BertPreTokenizer                       -> whitespace / punctuation
  CJK             BertNormalizer(handle_chinese_chars)   -> padded with spaces

Only the canonical combining class (needed to reorder the rare non-Mn combining
marks after decomposition) comes from Python's unicodedata, and the
alphanumeric set of the BM25 tokenizer (Rust char::is_alphanumeric =
Alphabetic property or general category N*; crates/kjarni-search/src/bm25.rs:191-197)
plus the Cased / Case_Ignorable sets behind str::to_lowercase's Final_Sigma rule
from the `regex` module's Unicode database.

Run:  python tools/gen_unicode_tables.py   (about a minute)
"""
import os
import sys
import unicodedata

import regex
from tokenizers import normalizers, pre_tokenizers

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "kjarni_amd", "csrc", "unicode_tables.inc")

clean = normalizers.BertNormalizer(clean_text=True, handle_chinese_chars=False, strip_accents=False,
                                   lowercase=False)
strip = normalizers.BertNormalizer(clean_text=False, handle_chinese_chars=False, strip_accents=True,
                                   lowercase=False)
cjk = normalizers.BertNormalizer(clean_text=False, handle_chinese_chars=True, strip_accents=False,
                                 lowercase=False)
nfd = normalizers.NFD()
lower = normalizers.Lowercase()
pre = pre_tokenizers.BertPreTokenizer()


def code_points():
    for cp in range(0x110000):
        if 0xD800 <= cp <= 0xDFFF:
            continue
        yield cp


def to_ranges(cps):
    cps = sorted(cps)
    out = []
    for cp in cps:
        if out and out[-1][1] + 1 == cp:
            out[-1][1] = cp
        else:
            out.append([cp, cp])
    return out


def main():
    drop, space, mn, ws, punct, cjk_set = [], [], [], [], [], []
    alnum = []
    alnum_re = regex.compile(r"[\p{Alphabetic}\p{N}]")
    cased, case_ign = [], []
    cased_re, case_ign_re = regex.compile(r"\p{Cased}"), regex.compile(r"\p{Case_Ignorable}")
    decomp, lowmap, ccc = {}, {}, {}
    S_BASE, L_BASE, V_BASE, T_BASE = 0xAC00, 0x1100, 0x1161, 0x11A7
    for cp in code_points():
        ch = chr(cp)
        c = clean.normalize_str(ch)
        if c == "":
            drop.append(cp)
        elif c == " " and ch != " ":
            space.append(cp)
        elif c != ch:
            raise SystemExit(f"unexpected clean_text result for U+{cp:04X}: {c!r}")
        d = nfd.normalize_str(ch)
        if d != ch:
            if S_BASE <= cp < S_BASE + 11172:
                si = cp - S_BASE
                exp = [L_BASE + si // 588, V_BASE + (si % 588) // 28]
                if si % 28:
                    exp.append(T_BASE + si % 28)
                assert [ord(x) for x in d] == exp, hex(cp)  # algorithmic in C++
            else:
                decomp[cp] = [ord(x) for x in d]
        else:
            if strip.normalize_str(ch) == "":
                mn.append(cp)
        lo = lower.normalize_str(ch)
        if lo != ch:
            lowmap[cp] = [ord(x) for x in lo]
        if cp not in (0,):
            parts = [p for p, _ in pre.pre_tokenize_str("a" + ch + "b")]
            if parts == ["a", "b"]:
                ws.append(cp)
            elif parts == ["a", ch, "b"]:
                punct.append(cp)
            elif parts != ["a" + ch + "b"]:
                raise SystemExit(f"unexpected pre-tokenizer split for U+{cp:04X}: {parts}")
        if cjk.normalize_str(ch) == " " + ch + " ":
            cjk_set.append(cp)
        if alnum_re.match(ch):
            alnum.append(cp)
        if cased_re.match(ch):
            cased.append(cp)
        if case_ign_re.match(ch):
            case_ign.append(cp)
        k = unicodedata.combining(ch)
        if k:
            ccc[cp] = k
    # every decomposition component that is dropped by strip_accents must be in mn (sanity)
    mn_set = set(mn)
    for cp, seq in decomp.items():
        kept = [x for x in seq if x not in mn_set]
        got = [ord(x) for x in strip.normalize_str(chr(cp))]
        assert kept == got, (hex(cp), kept, got)

    def emit_ranges(f, name, cps):
        r = to_ranges(cps)
        f.write(f"static const uint32_t {name}[][2] = {{\n")
        for i in range(0, len(r), 6):
            f.write("    " + " ".join(f"{{0x{a:X},0x{b:X}}}," for a, b in r[i:i + 6]) + "\n")
        f.write("};\n")
        f.write(f"static const size_t {name}_len = {len(r)};\n\n")

    def emit_map(f, name, m):
        pool = []
        f.write(f"static const UnicodeMapEntry {name}[] = {{\n")
        items = sorted(m.items())
        for i in range(0, len(items), 5):
            row = []
            for cp, seq in items[i:i + 5]:
                row.append(f"{{0x{cp:X},{len(pool)},{len(seq)}}},")
                pool.extend(seq)
            f.write("    " + " ".join(row) + "\n")
        f.write("};\n")
        f.write(f"static const size_t {name}_len = {len(items)};\n")
        f.write(f"static const uint32_t {name}_pool[] = {{\n")
        for i in range(0, len(pool), 12):
            f.write("    " + ",".join(f"0x{x:X}" for x in pool[i:i + 12]) + ",\n")
        f.write("};\n\n")

    with open(OUT, "w") as f:
        f.write("// GENERATED by tools/gen_unicode_tables.py -- do not edit.\n")
        f.write(f"// Probed from tokenizers {__import__('tokenizers').__version__} "
                f"(ccc from unicodedata {unicodedata.unidata_version}).\n\n")
        emit_ranges(f, "kCleanDrop", drop)
        emit_ranges(f, "kCleanSpace", space)
        emit_ranges(f, "kMarkNonspacing", mn)
        emit_ranges(f, "kWhitespace", ws)
        emit_ranges(f, "kPunctuation", punct)
        emit_ranges(f, "kCjk", cjk_set)
        emit_ranges(f, "kAlnum", alnum)
        emit_ranges(f, "kCased", cased)
        emit_ranges(f, "kCaseIgnorable", case_ign)
        emit_map(f, "kDecomp", decomp)
        emit_map(f, "kLower", lowmap)
        # ccc as ranges with value
        items = sorted(ccc.items())
        runs = []
        for cp, k in items:
            if runs and runs[-1][1] + 1 == cp and runs[-1][2] == k:
                runs[-1][1] = cp
            else:
                runs.append([cp, cp, k])
        f.write("static const uint32_t kCcc[][3] = {\n")
        for i in range(0, len(runs), 5):
            f.write("    " + " ".join(f"{{0x{a:X},0x{b:X},{k}}}," for a, b, k in runs[i:i + 5]) + "\n")
        f.write("};\n")
        f.write(f"static const size_t kCcc_len = {len(runs)};\n")
    print(f"wrote {OUT}: drop {len(drop)}, space {len(space)}, Mn {len(mn)}, ws {len(ws)}, "
          f"punct {len(punct)}, cjk {len(cjk_set)}, decomp {len(decomp)}, lower {len(lowmap)}, ccc {len(ccc)}")


if __name__ == "__main__":
    sys.exit(main())
