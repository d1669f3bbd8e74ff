"""The reference's two downstream tools over the files the hot path writes, same options, same
output:

    panfeed-get-clusters  /root/reference/panfeed/get_clusters.py:71-101
    panfeed-get-kmers     /root/reference/panfeed/get_kmers.py:88-145

What costs time in them is streaming `kmers_to_hashes.tsv` (one row per kept k-mer of the whole pangenome) and
`kmers.tsv` through pandas in 100 000-row chunks only to keep the few rows whose hash / cluster is in a set.  Here that
row filter runs on the GPU over the raw text (`RowFilter` -> pf_rowfilter_scan, csrc/pf_rowfilter.hip); the small
tables that remain (the associations, the kept rows) go through the same pandas statements as the reference's, so the
printed tables are the same bytes.  There is no CPU fallback for the filter.

One thing the reference leaves to chance is kept out of the comparison: it iterates over Python `set`s of cluster
names, so the order of its printed clusters / blocks changes with PYTHONHASHSEED.  Here clusters come in order of their
first appearance in kmers_to_hashes.tsv.
"""
import argparse
import ctypes as C
import gzip
import io
import logging
import sys

import pandas as pd

from . import _lib

logger = logging.getLogger("panfeed")

BLOCK_BYTES = 256 << 20


class RowFilter:
    """rows of a TSV whose first (`first_field=True`) or last field is one of `keys`, filtered on the device"""

    def __init__(self, keys, first_field, device=0):
        self.L = _lib.load()
        ks = [k.encode() if isinstance(k, str) else bytes(k) for k in keys]
        arr = (C.c_char_p * max(len(ks), 1))(*ks)
        lens = (C.c_uint32 * max(len(ks), 1))(*[len(k) for k in ks])
        self.h = C.c_void_p()
        _lib.check(self.L.pf_rowfilter_create(int(device), 1 if first_field else 0, arr, lens, len(ks), C.byref(self.h)))

    def close(self):
        if self.h:
            self.L.pf_rowfilter_destroy(self.h)
            self.h = C.c_void_p()

    __del__ = close

    def stats(self):
        n, ms = C.c_uint64(), C.c_float()
        _lib.check(self.L.pf_rowfilter_stats(self.h, C.byref(n), C.byref(ms)))
        return {"bytes_scanned": int(n.value), "device_ms": float(ms.value)}

    def scan_block(self, data, n=None):
        """(matching lines of the complete lines of `data` (bytes, or the first n bytes of a bytearray), joined; number of
        bytes consumed)"""
        b, e = C.POINTER(C.c_uint64)(), C.POINTER(C.c_uint64)()
        cnt, used = C.c_uint64(), C.c_uint64()
        if isinstance(data, bytearray):
            n = len(data) if n is None else n
            ptr = (C.c_char * len(data)).from_buffer(data)          # no copy
        else:
            n, ptr = len(data), data
        _lib.check(self.L.pf_rowfilter_scan(self.h, ptr, n, C.byref(b), C.byref(e), C.byref(cnt), C.byref(used)))
        view = memoryview(data)
        got = b"".join(view[b[i]:e[i]] for i in range(cnt.value))
        del ptr, view
        return got, int(used.value)

    def filter_file(self, path, block_bytes=None):
        """(header line, matching data lines) of a TSV file (.gz read through gzip, as pandas does by the name).  The file
        is read block by block straight into one buffer; what follows a block's last complete line is moved to the front
        for the next block."""
        block_bytes = block_bytes or BLOCK_BYTES
        opener = gzip.open if str(path).endswith(".gz") else open
        out = []
        with opener(path, "rb") as fh:
            header = fh.readline()
            buf = bytearray(block_bytes + (1 << 16))
            have = 0                                         # bytes carried over, at the front of buf
            while True:
                if have + block_bytes > len(buf):            # a line longer than the slack
                    buf.extend(bytes(have + block_bytes - len(buf)))
                got_n = fh.readinto(memoryview(buf)[have:have + block_bytes])
                if not got_n:
                    break
                total = have + got_n
                got, used = self.scan_block(buf, total)
                out.append(got)
                have = total - used
                buf[:have] = buf[used:total]
            if have:                                         # a last line without its newline
                got, _ = self.scan_block(bytes(buf[:have]) + b"\n")
                out.append(got)
        return header, b"".join(out)


def _table(header, rows):
    return pd.read_csv(io.BytesIO(header + rows), sep="\t")


def _options(description, kmers):
    p = argparse.ArgumentParser(description=description)
    p.add_argument("-a", "--associations", required=True)
    p.add_argument("-p", "--kmers-to-hashes", required=True)
    if kmers:
        p.add_argument("-k", "--kmers", required=True)
    p.add_argument("-t", "--threshold", type=float, default=1)
    p.add_argument("-c", "--column", default="lrt-pvalue")
    p.add_argument("-o", "--output", default=None)
    if kmers:
        p.add_argument("--only-passing", action="store_true", default=False)
        p.add_argument("--clusters-per-iteration", type=int, default=15)
    p.add_argument("-v", action="count", default=0)
    p.add_argument("--device", type=int, default=0, help="GPU the row filter runs on")
    return p


def _associations(args, index_name=None):
    """the filtered associations table and the passing hashes (get_clusters.py:76-88, get_kmers.py:93-106)"""
    a = pd.read_csv(args.associations, sep="\t", index_col=0)
    if index_name:
        a.index.name = index_name
    if args.column not in a.columns:
        logger.warning(f"Associations file does not have the {args.column} column")
        sys.exit(1)
    a = a[a[args.column] <= args.threshold]
    if args.output is not None:
        a.to_csv(args.output, sep="\t")
    return a, [str(x) for x in a.index.unique()]


_NAN_KEY = float("nan")      # ONE object for every NaN a column holds (tolist() makes a new one per cell, and nan != nan)


def _key(v):
    return _NAN_KEY if isinstance(v, float) and v != v else v


def _ordered_unique(series):
    return list(dict.fromkeys(_key(v) for v in series.tolist()))


def _first_fields(rows):
    """the literal first field of every line of `rows` (bytes), in order"""
    return [ln.split(b"\t", 1)[0] for ln in rows.split(b"\n") if ln]


def get_clusters(argv=None, out=None):
    """panfeed-get-clusters: the gene clusters that have a k-mer whose pattern passes the threshold, one per line"""
    out = out or sys.stdout
    args = _options("Indicate which genes clusters have significantly associated patterns", False).parse_args(argv)
    a, passing = _associations(args)
    f = RowFilter(passing, first_field=False, device=args.device)
    try:
        header, rows = f.filter_file(args.kmers_to_hashes)
    finally:
        f.close()
    h = _table(header, rows)
    for c in _ordered_unique(h["cluster"]):
        print(c, file=out)
    return 0


def get_kmers(argv=None, out=None):
    """panfeed-get-kmers: association results joined with the k-mers' clusters and positions"""
    out = out or sys.stdout
    args = _options("Annotate association results with positional information", True).parse_args(argv)
    a, passing = _associations(args, index_name="hashed_pattern")
    f = RowFilter(passing, first_field=False, device=args.device)
    try:
        header, rows = f.filter_file(args.kmers_to_hashes)
    finally:
        f.close()
    h = _table(header, rows).set_index("hashed_pattern")
    clusters = _ordered_unique(h["cluster"])
    # The device filter compares the TEXT of a row's cluster field, the reference the values pandas parsed on both sides
    # (get_kmers.py:131-134): a cluster named '007' or '1e3' parses as a number whose str() is not the file's bytes.  The
    # keys given to the filter are therefore the literal fields of the kept kmers_to_hashes rows, grouped by the value
    # pandas made of them (row i of `h` is line i of `rows`).
    # (a value pandas read as NaN -- numeric cluster ids plus an 'NA' -- is a different object at every look: _key)
    literal = {}
    for val, lit in zip(h["cluster"].tolist(), _first_fields(rows)):
        literal.setdefault(_key(val), {})[lit] = None
    first = True
    b = a.join(h, how="inner") if clusters else None                       # get_kmers.py:136
    for idx in range(0, len(clusters), args.clusters_per_iteration):
        bunch = clusters[idx: idx + args.clusters_per_iteration]
        # (a NaN among the bunch selects nothing: the reference's `x['cluster'].isin(bunch)`, get_kmers.py:131-134, is False
        # for a NaN cell when the bunch is a list of the column's unique() values -- such rows are dropped, not matched)
        fk = RowFilter([lit for c in bunch if c is not _NAN_KEY for lit in literal[_key(c)]], first_field=True, device=args.device)
        try:
            kheader, krows = fk.filter_file(args.kmers)
        finally:
            fk.close()
        k = _table(kheader, krows).set_index(["cluster", "k-mer"])
        how = "left" if args.only_passing else "right"                      # get_kmers.py:137-141
        t = b.reset_index().set_index(["cluster", "k-mer"]).join(k, how=how)
        t.to_csv(out, sep="\t", header=first)
        first = False
    return 0


def main_get_clusters():
    sys.exit(get_clusters())


def main_get_kmers():
    sys.exit(get_kmers())
