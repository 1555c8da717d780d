"""What BASELINE.md section 3 asks to be logged for every generated index / parity sample: the duplicate-code rate and the number of
parity queries whose top-(k + 1) holds an exact fp64 tie (at the k / k + 1 boundary: membership would depend on the LingPipe
assumption A1; anywhere inside: the order of the tied ids would).  Used by bench.py, tests/bench_configs.py and
tests/test_queue_rules_cpu.py; numpy only."""
import numpy as np


def tie_census(d_k1, k):
    """d_k1: [nq][k + 1] ascending distances (inf-padded) of the top-(k + 1)"""
    d = np.asarray(d_k1)
    fin = np.isfinite(d)
    full = d.shape[1] > k
    at_k = int(np.sum((d[:, k - 1] == d[:, k]) & fin[:, k])) if full else 0
    any_tie = int(np.sum(np.any((d[:, 1:] == d[:, :-1]) & fin[:, 1:], axis=1)))
    return {"queries": int(d.shape[0]), "queries_with_tie_at_k": at_k, "queries_with_any_tie_in_top_k_plus_1": any_tie}


def duplicate_code_rate(off, codes, lists=None):
    """fraction of the stored codes that repeat an earlier code of their own list; `lists`: the sample of lists to look at (all)"""
    off = np.asarray(off)
    lists = range(len(off) - 1) if lists is None else lists
    dup = tot = 0
    for c in lists:
        blk = np.ascontiguousarray(codes[off[c]:off[c + 1]])
        tot += len(blk)
        if len(blk):
            dup += len(blk) - len(np.unique(blk.view([("", blk.dtype)] * blk.shape[1])))
    return dup / max(1, tot), int(tot)
