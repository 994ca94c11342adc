"""CPU restatement of the retrieval side: BM25, reciprocal-rank fusion, metadata
filter, the on-disk segment format and IndexReader's search functions.

TEST INFRASTRUCTURE ONLY (tests/ imports it; the product never does).
Pure Python (small cases).  Paths below are relative to /root/reference/crates/.

Pinned by the reference's own unit tests, reproduced in tests/test_search_host.py:
kjarni-search/src/bm25.rs:200-561, kjarni-search/src/hybrid.rs:35-62,
kjarni-rag/src/index_reader.rs:355-1048 (MetadataFilter cases),
kjarni-rag/src/segment.rs:378-434 (segment round trip).
"""
from __future__ import annotations

import json
import math
import os
import struct
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

F32 = np.float32


# ----------------------------------------------------------------------------- BM25
def tokenize(text: str) -> List[str]:
    """kjarni-search/src/bm25.rs:191-197: lowercase, split on non-alphanumeric
    chars, keep pieces of at least 2 BYTES."""
    out, cur = [], []
    for ch in text.lower():
        if ch.isalnum():
            cur.append(ch)
        else:
            if cur:
                out.append("".join(cur))
                cur = []
    if cur:
        out.append("".join(cur))
    return [t for t in out if len(t.encode("utf-8")) >= 2]


class Bm25Index:
    """kjarni-search/src/bm25.rs:42-189.  f32 arithmetic as in the reference."""

    def __init__(self):
        self.doc_frequencies: Dict[str, int] = {}
        self.doc_lengths: List[int] = []
        self.avg_doc_length = F32(0.0)
        self.total_docs = 0
        self.inverted_index: Dict[str, List[Tuple[int, int]]] = {}
        self.k1, self.b, self.epsilon = F32(1.2), F32(0.75), F32(0.25)
        self.total_length = 0

    def add_document(self, doc_id: int, text: str):  # bm25.rs:108-139
        tokens = tokenize(text)
        if doc_id >= len(self.doc_lengths):
            self.doc_lengths.extend([0] * (doc_id + 1 - len(self.doc_lengths)))
        self.doc_lengths[doc_id] = len(tokens)
        counts: Dict[str, int] = {}
        for t in tokens:
            counts[t] = counts.get(t, 0) + 1
        for term, c in counts.items():
            self.inverted_index.setdefault(term, []).append((doc_id, c))
            self.doc_frequencies[term] = self.doc_frequencies.get(term, 0) + 1
        self.total_docs = max(self.total_docs, doc_id + 1)
        self.total_length += len(tokens)
        self.avg_doc_length = F32(self.total_length) / F32(self.total_docs)

    def term_frequency(self, term: str, doc_id: int) -> int:  # bm25.rs:177-188
        for d, c in self.inverted_index.get(term, []):
            if d == doc_id:
                return c
        return 0

    def score(self, query_tokens: Sequence[str], doc_id: int) -> np.float32:  # bm25.rs:141-175
        score = F32(0.0)
        doc_length = F32(self.doc_lengths[doc_id])
        length_norm = F32(1.0) - self.b + self.b * (doc_length / self.avg_doc_length)
        for term in query_tokens:
            tf = F32(self.term_frequency(term, doc_id))
            if tf == 0:
                continue
            df = F32(self.doc_frequencies.get(term, 0))
            if df == 0:
                continue
            idf = F32(math.log(float((F32(self.total_docs) - df + F32(0.5)) / (df + F32(0.5)) + F32(1.0))))
            ntf = (tf * (self.k1 + F32(1.0))) / (tf + self.k1 * length_norm)
            score = F32(score + idf * ntf)
        return score

    def search(self, query: str, limit: int) -> List[Tuple[int, float]]:  # bm25.rs:84-107
        if self.total_docs == 0:
            return []
        q = tokenize(query)
        if not q:
            return []
        res = []
        for d in range(self.total_docs):
            s = self.score(q, d)
            if s > 0:
                res.append((d, float(s)))
        # the reference sorts a HashMap's entries: ties have no defined order; use ascending id
        res.sort(key=lambda x: (-x[1], x[0]))
        return res[:limit]

    # bincode 1.x (fixed-width little-endian ints, u64 lengths) of the struct, field order as declared
    def to_bincode(self) -> bytes:
        def s(x: str) -> bytes:
            b = x.encode("utf-8")
            return struct.pack("<Q", len(b)) + b
        out = [struct.pack("<Q", len(self.doc_frequencies))]
        for k, v in self.doc_frequencies.items():
            out += [s(k), struct.pack("<Q", v)]
        out.append(struct.pack("<Q", len(self.doc_lengths)))
        out += [struct.pack("<Q", v) for v in self.doc_lengths]
        out += [struct.pack("<f", float(self.avg_doc_length)), struct.pack("<Q", self.total_docs)]
        out.append(struct.pack("<Q", len(self.inverted_index)))
        for k, postings in self.inverted_index.items():
            out += [s(k), struct.pack("<Q", len(postings))]
            out += [struct.pack("<QQ", d, c) for d, c in postings]
        out.append(struct.pack("<fff", float(self.k1), float(self.b), float(self.epsilon)))
        out.append(struct.pack("<Q", 0))  # token_to_docs: never populated by add_document
        out.append(struct.pack("<Q", self.total_length))
        return b"".join(out)


# ----------------------------------------------------------------------------- RRF
def hybrid_search(keyword: Sequence[Tuple[int, float]], semantic: Sequence[Tuple[int, float]], limit: int):
    """kjarni-search/src/hybrid.rs:3-31: reciprocal-rank fusion, k = 60."""
    comb: Dict[int, np.float32] = {}
    for lst in (keyword, semantic):
        for rank, (idx, _) in enumerate(lst):
            comb[idx] = F32(comb.get(idx, F32(0.0)) + F32(1.0) / (F32(60.0) + F32(rank + 1)))
    res = sorted(comb.items(), key=lambda x: (-x[1], x[0]))
    return [(i, float(s)) for i, s in res[:limit]]


# ----------------------------------------------------------------------------- filter
def glob_match(pattern: str, path: str) -> bool:
    """glob-match 0.2.1 (Cargo.lock; called at kjarni-rag/src/index_reader.rs:63-69), restated as a
    translation to a regular expression over bytes: leading `!` negates; `\\x` is a literal x;
    `?` one non-'/' byte (the crate matches over &[u8]); `[a-z]`/`[!a]` byte classes; `*` stays inside a path component;
    `**` as a whole segment crosses components (and `**/` may match nothing), a trailing `**`
    crosses components, any other `**` acts as `*`; `{a,b}` alternatives."""
    import re

    class Invalid(Exception):
        pass

    pat = pattern.encode("utf-8")
    neg = False
    k = 0
    while k < len(pat) and pat[k:k + 1] == b"!":
        neg = not neg
        k += 1

    def lit(b: int) -> bytes:
        return re.escape(bytes([b]))

    def conv(p: bytes, lead: bytes, is_tail: bool) -> bytes:
        # `lead`: the pattern byte just before p (b"" at the true start); `is_tail`: p ends the pattern
        i, out = 0, []
        n = len(p)
        while i < n:
            c = p[i:i + 1]
            if c == b"*":
                if p[i:i + 2] == b"**":
                    left_ok = (lead == b"") if i == 0 else p[i - 1:i] == b"/"
                    j = i + 2
                    while left_ok and p[j:j + 3] == b"/**" and (j + 3 == n or p[j + 3:j + 4] == b"/"):
                        j += 3
                    if j == n and is_tail:
                        out.append(b".*")
                    elif left_ok and p[j:j + 1] == b"/":
                        out.append(b"(?:.*/)?")
                        j += 1
                    else:
                        out.append(b"[^/]*")
                    i = j
                    continue
                out.append(b"[^/]*")
                i += 1
            elif c == b"?":
                out.append(b"[^/]")
                i += 1
            elif c == b"[":
                i += 1
                negc = p[i:i + 1] in (b"^", b"!")
                if negc:
                    i += 1
                items, first = [], True
                while i < n and (first or p[i:i + 1] != b"]"):
                    def take():
                        nonlocal i
                        if p[i:i + 1] == b"\\":
                            if i + 1 >= n:
                                raise Invalid()
                            i += 1
                        v = p[i]
                        i += 1
                        return v
                    lo = take()
                    if i + 1 < n and p[i:i + 1] == b"-" and p[i + 1:i + 2] != b"]":
                        i += 1
                        hi = take()
                    else:
                        hi = lo
                    items.append((lo, hi))
                    first = False
                if i >= n:
                    raise Invalid()
                i += 1
                allowed = [b for b in range(256) if any(lo <= b <= hi for lo, hi in items) != negc]
                out.append(b"(?:" + b"|".join(lit(b) for b in allowed) + b")" if allowed else b"(?!)")
            elif c == b"{":
                depth, j, cuts, close = 0, i, [], None
                while j < n:
                    d = p[j:j + 1]
                    if d == b"\\":
                        j += 2
                        continue
                    if d == b"[":
                        r = j + 1
                        if p[r:r + 1] in (b"!", b"^"):
                            r += 1
                        if p[r:r + 1] == b"]":
                            r += 1
                        while r < n and p[r:r + 1] != b"]":
                            r += 2 if p[r:r + 1] == b"\\" else 1
                        j = r + 1
                        continue
                    if d == b"{":
                        depth += 1
                    elif d == b"}":
                        depth -= 1
                        if depth == 0:
                            close = j
                            break
                    elif d == b"," and depth == 1:
                        cuts.append(j)
                    j += 1
                if close is None:
                    raise Invalid()
                cuts.append(close)
                alts, a0 = [], i + 1
                for cut in cuts:
                    alts.append(conv(p[a0:cut], p[a0 - 1:a0], False))
                    a0 = cut + 1
                # what follows the brace is converted knowing a '}' precedes it
                return b"".join(out) + b"(?:" + b"|".join(alts) + b")" + conv(p[close + 1:], b"}", is_tail)
            else:
                if c == b"\\":
                    if i + 1 >= n:
                        raise Invalid()
                    i += 1
                out.append(lit(p[i]))
                i += 1
        return b"".join(out)

    try:
        rx = conv(pat[k:], b"" if k == 0 else b"!", True)
    except Invalid:
        return neg
    return (re.fullmatch(rx, path.encode("utf-8"), flags=re.S) is not None) != neg


class MetadataFilter:
    """kjarni-rag/src/index_reader.rs:13-101."""

    def __init__(self):
        self.must_match: Dict[str, str] = {}
        self.must_not_match: Dict[str, str] = {}
        self.source_patterns: List[str] = []

    def must(self, k, v):
        self.must_match[k] = v
        return self

    def must_not(self, k, v):
        self.must_not_match[k] = v
        return self

    def source(self, p):
        self.source_patterns.append(p)
        return self

    def matches(self, md: Dict[str, str]) -> bool:
        for k, v in self.must_match.items():
            if md.get(k) != v:
                return False
        for k, v in self.must_not_match.items():
            if md.get(k) == v:
                return False
        if self.source_patterns:
            src = md.get("source")
            if src is None:
                return False
            fname = os.path.basename(src) or src
            if not any(glob_match(p, src if "/" in p else fname) for p in self.source_patterns):
                return False
        return True


# ----------------------------------------------------------------------------- on-disk index
def write_index(root: str, dimension: int, docs: Sequence[Tuple[str, np.ndarray, Dict[str, str]]],
                max_docs_per_segment: int = 10_000, model_name: Optional[str] = None):
    """Writes the reference's index layout: <root>/config.json (kjarni-rag/src/config.rs:5-13)
    and <root>/segments/<id>/{vectors.bin, docs.bin, docs.idx, bm25.bin, metadata.jsonl,
    segment.json} (kjarni-rag/src/segment.rs:84-197)."""
    os.makedirs(os.path.join(root, "segments"), exist_ok=True)
    cfg = dict(dimension=dimension, max_docs_per_segment=max_docs_per_segment,
               max_segment_memory=100 * 1024 * 1024, embedding_model=None, model_name=model_name,
               created_at=0, version=1)
    with open(os.path.join(root, "config.json"), "w") as f:
        json.dump(cfg, f)
    seg_id = 0
    for s0 in range(0, len(docs), max_docs_per_segment):
        chunk = docs[s0:s0 + max_docs_per_segment]
        d = os.path.join(root, "segments", f"seg_{seg_id:06d}")
        os.makedirs(d, exist_ok=True)
        bm = Bm25Index()
        offsets, cur = [], 0
        with open(os.path.join(d, "vectors.bin"), "wb") as fv, open(os.path.join(d, "docs.bin"), "wb") as fd, \
                open(os.path.join(d, "metadata.jsonl"), "w") as fm:
            for i, (text, emb, md) in enumerate(chunk):
                fv.write(np.ascontiguousarray(emb, dtype="<f4").tobytes())
                offsets.append(cur)
                tb = text.encode("utf-8")
                fd.write(tb + b"\n")
                cur += len(tb) + 1
                fm.write(json.dumps(md or {}) + "\n")
                bm.add_document(i, text)
        with open(os.path.join(d, "docs.idx"), "wb") as f:
            f.write(struct.pack("<Q", len(offsets)) + b"".join(struct.pack("<Q", o) for o in offsets))
        with open(os.path.join(d, "bm25.bin"), "wb") as f:
            f.write(bm.to_bincode())
        vs = os.path.getsize(os.path.join(d, "vectors.bin"))
        ds = os.path.getsize(os.path.join(d, "docs.bin"))
        with open(os.path.join(d, "segment.json"), "w") as f:
            json.dump(dict(id=seg_id, doc_count=len(chunk), dimension=dimension, created_at=0,
                           total_bytes=vs + ds), f)
        seg_id += 1


class IndexOracle:
    """IndexReader restated over in-memory documents (kjarni-rag/src/index_reader.rs:160-330);
    segments of `max_docs_per_segment` documents, global id = position."""

    def __init__(self, docs, max_docs_per_segment=10_000):
        self.segs = [docs[i:i + max_docs_per_segment] for i in range(0, len(docs), max_docs_per_segment)]
        self.offsets = np.cumsum([0] + [len(s) for s in self.segs]).tolist()
        self.bm = []
        for s in self.segs:
            b = Bm25Index()
            for i, (t, _, _) in enumerate(s):
                b.add_document(i, t)
            self.bm.append(b)

    def _result(self, seg, loc, score):
        t, _, md = self.segs[seg][loc]
        return dict(score=float(score), document_id=self.offsets[seg] + loc, text=t, metadata=dict(md or {}))

    def search_semantic(self, q, limit):
        from oracle import oracle as O
        allr = []
        for si, s in enumerate(self.segs):
            corpus = np.stack([np.asarray(e, np.float32) for _, e, _ in s])
            idx, sc = O.search(np.asarray(q, np.float32), corpus, limit, mode=1)
            allr += [(si, int(i), float(v)) for i, v in zip(idx, sc)]
        allr.sort(key=lambda x: -x[2])  # stable: segment order, then the segment's own order
        return [self._result(*r) for r in allr[:limit]]

    def search_keywords(self, query, limit):
        allr = []
        for si, b in enumerate(self.bm):
            allr += [(si, d, s) for d, s in b.search(query, limit)]
        allr.sort(key=lambda x: -x[2])
        return [self._result(*r) for r in allr[:limit]]

    def search_hybrid(self, query, q, limit):
        kw = [(r["document_id"], r["score"]) for r in self.search_keywords(query, limit * 2)]
        sem = [(r["document_id"], r["score"]) for r in self.search_semantic(q, limit * 2)]
        out = []
        for gid, score in hybrid_search(kw, sem, limit):
            seg = max(i for i, o in enumerate(self.offsets[:-1]) if o <= gid)
            r = self._result(seg, gid - self.offsets[seg], score)
            out.append(r)
        return out


# ----------------------------------------------------------------------------- write side: chunking + discovery
class TextSplitter:
    """kjarni-rag/src/splitter.rs:44-170.  `len()` of a Rust str is BYTES (sections and the running
    chunk are measured in bytes); oversized sections and the overlap suffix are cut in CHARACTERS."""

    def __init__(self, chunk_size=1000, chunk_overlap=200, separator="\n\n"):
        if chunk_size == 0:
            raise ValueError("chunk_size must be greater than 0")
        if chunk_overlap >= chunk_size:
            raise ValueError("chunk_overlap must be less than chunk_size")
        self.chunk_size, self.chunk_overlap, self.separator = chunk_size, chunk_overlap, separator

    @staticmethod
    def _blen(s: str) -> int:
        return len(s.encode("utf-8"))

    def _split_large(self, text: str) -> List[str]:  # splitter.rs:132-165
        out, start, n = [], 0, len(text)
        while start < n:
            end = min(start + self.chunk_size, n)
            out.append(text[start:end])
            if end >= n:
                break
            start += self.chunk_size - self.chunk_overlap if 0 < self.chunk_overlap < self.chunk_size else self.chunk_size
        return out

    def split(self, text: str) -> List[str]:  # splitter.rs:68-120
        if not text:
            return []
        chunks, cur = [], ""
        for section in text.split(self.separator):
            if not section:
                continue
            if self._blen(section) > self.chunk_size:
                if cur:
                    chunks.append(cur)
                    cur = ""
                chunks += self._split_large(section)
                continue
            would_be = self._blen(section) if not cur else self._blen(cur) + self._blen(self.separator) + self._blen(section)
            if would_be > self.chunk_size and cur:
                chunks.append(cur)
                if self.chunk_overlap > 0:
                    cur = cur if len(cur) <= self.chunk_overlap else cur[len(cur) - self.chunk_overlap:]
                else:
                    cur = ""
            if cur:
                cur += self.separator
            cur += section
        if cur:
            chunks.append(cur)
        return chunks


TEXT_EXTENSIONS = ("txt md markdown rst org json yaml yml toml xml csv html htm css rs py js ts go java c cpp h hpp "
                   "cs rb sh bash zsh fish ps1 sql r scala kt swift m mm lua pl php ex exs clj hs").split()


def is_supported_file(path: str, extensions: Sequence[str] = ()) -> bool:
    """loader.rs:181-198 / indexer/model.rs:797-813 (Path::extension, lowercased)."""
    name = os.path.basename(path.rstrip("/"))
    if "." not in name or name.rfind(".") == 0:
        return False
    ext = name[name.rfind(".") + 1:].lower()
    return ext in (extensions if extensions else TEXT_EXTENSIONS)


def collect_files(inputs: Sequence[str], recursive=True, include_hidden=False, extensions: Sequence[str] = (),
                  exclude_patterns: Sequence[str] = (), max_file_size: Optional[int] = 10 * 1024 * 1024) -> List[str]:
    """indexer/model.rs:727-795; directory entries in byte order of their names (the reference
    inherits the OS's readdir order)."""
    files: List[str] = []

    def walk(d: str):
        for name in sorted(os.listdir(d), key=lambda s: s.encode("utf-8", "surrogateescape")):
            p = d + name if d.endswith("/") else d + "/" + name
            if os.path.isdir(p) and not os.path.islink(p):
                if recursive:
                    walk(p)
                continue
            if not os.path.isfile(p):
                continue
            if not include_hidden and name.startswith("."):
                continue
            if any(glob_match(pat, p) for pat in exclude_patterns):
                continue
            if max_file_size is not None and os.path.getsize(p) > max_file_size:
                continue
            if is_supported_file(p, extensions):
                files.append(p)

    for inp in inputs:
        if not os.path.exists(inp):
            raise FileNotFoundError(inp)
        if os.path.isfile(inp):
            if is_supported_file(inp, extensions):
                files.append(inp)
        elif os.path.isdir(inp):
            walk(inp)
    return files


def load_file_chunks(path: str, splitter: TextSplitter) -> List[Tuple[str, Dict[str, str]]]:
    """loader.rs:85-110 + ChunkMetadata::to_hashmap (kjarni-search/src/types.rs:58-77)."""
    with open(path, "rb") as f:
        content = f.read().decode("utf-8")  # fs::read_to_string: invalid UTF-8 is an error
    texts = splitter.split(content)
    return [(t, {"source": path, "chunk_index": str(i), "total_chunks": str(len(texts))}) for i, t in enumerate(texts)]


# ----------------------------------------------------------------------------- independent reader of the on-disk format
def _bincode_reader(blob: bytes):
    pos = [0]

    def u64():
        v, = struct.unpack_from("<Q", blob, pos[0])
        pos[0] += 8
        return v

    def f32():
        v, = struct.unpack_from("<f", blob, pos[0])
        pos[0] += 4
        return v

    def s():
        n = u64()
        v = blob[pos[0]:pos[0] + n].decode("utf-8")
        assert len(blob) >= pos[0] + n
        pos[0] += n
        return v
    return u64, f32, s, pos


def bm25_from_bincode(blob: bytes) -> "Bm25Index":
    """bincode 1.x image of kjarni-search/src/bm25.rs:42-60, field by field."""
    u64, f32, s, pos = _bincode_reader(blob)
    b = Bm25Index()
    for _ in range(u64()):
        k = s()
        b.doc_frequencies[k] = u64()
    b.doc_lengths = [u64() for _ in range(u64())]
    b.avg_doc_length = F32(f32())
    b.total_docs = u64()
    for _ in range(u64()):
        k = s()
        b.inverted_index[k] = [(u64(), u64()) for _ in range(u64())]
    b.k1, b.b, b.epsilon = F32(f32()), F32(f32()), F32(f32())
    for _ in range(u64()):  # token_to_docs
        s()
        for _ in range(u64()):
            u64()
    b.total_length = u64()
    assert pos[0] == len(blob), "trailing bytes in bm25.bin"
    return b


def read_index(root: str) -> Dict:
    """Everything an index directory holds, parsed without the library: config.json, index.json and,
    per segment, segment.json / vectors / texts / metadata / BM25."""
    out = dict(config=json.load(open(os.path.join(root, "config.json"))), segments=[], names=[])
    ip = os.path.join(root, "index.json")
    out["index"] = json.load(open(ip)) if os.path.exists(ip) else None
    segdir = os.path.join(root, "segments")
    for name in sorted(os.listdir(segdir)):
        d = os.path.join(segdir, name)
        meta = json.load(open(os.path.join(d, "segment.json")))
        n, dim = meta["doc_count"], meta["dimension"]
        vec = np.fromfile(os.path.join(d, "vectors.bin"), dtype="<f4").reshape(n, dim)
        blob = open(os.path.join(d, "docs.idx"), "rb").read()
        cnt, = struct.unpack_from("<Q", blob, 0)
        offs = list(struct.unpack_from(f"<{cnt}Q", blob, 8))
        assert len(blob) == 8 + 8 * cnt and cnt == n
        docs = open(os.path.join(d, "docs.bin"), "rb").read()
        ends = offs[1:] + [len(docs)]
        texts = [docs[a:b - 1].decode("utf-8") for a, b in zip(offs, ends)]
        assert all(docs[b - 1:b] == b"\n" for b in ends)
        mds = [json.loads(line) for line in open(os.path.join(d, "metadata.jsonl"), encoding="utf-8").read().split("\n")[:-1]]
        assert len(mds) == n
        bm = bm25_from_bincode(open(os.path.join(d, "bm25.bin"), "rb").read())
        out["segments"].append(dict(meta=meta, vectors=vec, texts=texts, metadata=mds, bm25=bm))
        out["names"].append(name)
    return out
