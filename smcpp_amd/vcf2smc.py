"""VCF -> SMC++ row format (the reference's `smcpp/commands/vcf2smc.py:73-271`, which needs pysam).

Plain-text VCF reader (gzip or not), one contig, bi-allelic single-base records only; semantics restated from the
reference: the distinguished pair is the first allele of `d[0]` and the second allele of `d[1]` (default: both alleles
of the first sample of population 1), every other allele of the listed samples is undistinguished; sites where the whole
sub-sample is derived are folded to non-segregating; gaps between records are non-segregating runs (or missing, beyond
`missing_cutoff`); consecutive equal observations are merged as `util.RepeatingWriter` does.  No BED mask support.
"""
from __future__ import annotations

import gzip
import json
from typing import List, Optional, Sequence, Tuple

import numpy as np

from .data import Contig


def _open(path):
    with open(path, "rb") as f:
        magic = f.read(2)
    return gzip.open(path, "rt") if magic == b"\x1f\x8b" else open(path, "rt")


def vcf2smc(vcf: str, contig: str, pop1: Tuple[str, Sequence[str]], pop2: Optional[Tuple[str, Sequence[str]]] = None,
            d: Optional[Sequence[str]] = None, length: Optional[int] = None, missing_cutoff: Optional[int] = None,
            ignore_missing: bool = False, drop_first_last: bool = False):
    """Returns `(Contig, header_dict)`; `Contig.data` holds the int32 rows `[span, (a, b, nb) per population]`."""
    pops = [pop1] + ([pop2] if pop2 is not None and pop2[0] is not None else [])
    for pid, ss in pops:
        if len(set(ss)) != len(ss):
            raise RuntimeError("Population %s has duplicated samples" % pid)
    if len(pops) == 2 and set(pops[0][1]) & set(pops[1][1]):
        raise RuntimeError("Populations 1 and 2 should be disjoint")
    if not d:
        d = [pop1[1][0]] * 2
    dtag = [(d[0], 0), (d[1], 1)]
    all_samples = set(s for _, ss in pops for s in ss)
    dist: List[List[Tuple[str, int]]] = [[] for _ in pops]
    for sid, i in dtag:
        if sid not in all_samples:
            raise RuntimeError("%s is not in the sample list" % sid)
        dist[0 if sid in pops[0][1] else 1].append((sid, i))
    undist = [[(k, i) for k in ss for i in (0, 1) if (k, i) not in dd] for (_, ss), dd in zip(pops, dist)]
    cutoff = np.inf if missing_cutoff is None else missing_cutoff
    contig_length = length
    samples: List[str] = []
    rows: List[List[int]] = []
    last_ob: Optional[List[int]] = None
    first = [True]

    def emit(ob):                                    # util.RepeatingWriter + the drop_first_last switch
        nonlocal last_ob
        if first[0] and drop_first_last:
            first[0] = False
            return
        first[0] = False
        if last_ob is None:
            last_ob = list(ob)
        elif ob[1:] == last_ob[1:]:
            last_ob[0] += ob[0]
        else:
            if last_ob[0] > 0:
                rows.append(last_ob)
            last_ob = list(ob)

    na = [len(x) for x in dist]
    col = {}
    last_pos = 0
    multiples = 0
    miss_row: List[int] = []
    nonseg_row: List[int] = []
    with _open(vcf) as f:
        for line in f:
            if line.startswith("##"):
                if line.startswith("##contig=<") and contig_length is None:
                    body = line.strip()[len("##contig=<"):-1]
                    kv = dict(x.split("=", 1) for x in body.split(",") if "=" in x)
                    if kv.get("ID") == contig and "length" in kv:
                        contig_length = int(kv["length"])
                continue
            if line.startswith("#CHROM"):
                samples = line.rstrip("\n").split("\t")[9:]
                col = {s: 9 + i for i, s in enumerate(samples)}
                if not set(dd[0] for dl in dist for dd in dl) <= set(samples):
                    raise RuntimeError("Distinguished lineages not found in data?")
                missing = [s for u in undist for s, _ in u if s not in samples]
                if missing and not ignore_missing:
                    raise RuntimeError("The following samples were not found in the data: %s. If you want to continue "
                                       "without these samples, use --ignore-missing." % ", ".join(missing))
                undist = [[t for t in u if t[0] not in missing] for u in undist]
                nbf = [len(u) for u in undist]
                miss_row = [-1, 0, 0] * len(nbf)
                nonseg_row = sum([[0, 0, x] for x in nbf], [])
                continue
            fld = line.rstrip("\n").split("\t")
            if fld[0] != contig:
                continue
            alleles = [fld[3]] + ([] if fld[4] == "." else fld[4].split(","))
            if len(alleles) > 2 or any(len(x) != 1 for x in alleles):
                continue                                                    # SNPs only
            pos = int(fld[1])
            gti = fld[8].split(":").index("GT")

            def gt(sid):
                g = fld[col[sid]].split(":")[gti].replace("|", "/").split("/")
                return [None if x == "." else alleles[int(x)] for x in g]

            ref = alleles[0]
            for dl in dist:
                for sid, _ in dl:
                    if len(gt(sid)) != 2:
                        raise RuntimeError("Expected a diploid genotype at position %d for individual %s" % (pos, sid))
            da = [[gt(sid)[i] for sid, i in dl] for dl in dist]
            a = [sum(x != ref for x in dd) if None not in dd else -1 for dd in da]
            bs = [[gt(sid)[i] != ref for sid, i in un if gt(sid)[i] is not None] for un in undist]
            b = [sum(x) for x in bs]
            nb = [len(x) for x in bs]
            if b == nb and a == na:                                         # whole sub-sample derived: fold
                a = [0] * len(a)
                b = [0] * len(b)
            abnb = [x for t in zip(a, b, nb) for x in t]
            if pos == last_pos:
                multiples += 1
                continue
            span = pos - last_pos - 1
            if 1 <= span <= cutoff:
                emit([span] + nonseg_row)
            elif span > cutoff:
                emit([span] + miss_row)
            emit([1] + abnb)
            last_pos = pos
    if contig_length is None:
        raise RuntimeError("Failed to acquire contig length from VCF header. See the length option.")
    if not drop_first_last:
        emit([contig_length - last_pos] + nonseg_row)
    if last_ob is not None and last_ob[0] > 0:
        rows.append(last_ob)
    header = {"version": "smcpp_amd", "pids": [p[0] for p in pops], "undist": [[list(t) for t in u] for u in undist],
              "dist": [[list(t) for t in dl] for dl in dist]}
    data = np.array(rows, dtype=np.int32).reshape(-1, 1 + 3 * len(pops))
    c = Contig(data=data, pid=tuple(p[0] for p in pops), n=[len(u) for u in undist], a=na, fn=vcf)
    c.multiples = multiples
    return c, header


def write_smc(path: str, contig: Contig, header: dict):
    """The `.smc[.gz]` text format `load_smc` reads back (`# SMC++ {json}` + one row of ints per line)."""
    op = gzip.open if path.endswith(".gz") else open
    with op(path, "wt") as f:
        f.write("# SMC++ " + json.dumps(header) + "\n")
        for r in contig.data:
            f.write(" ".join(str(int(x)) for x in r) + "\n")
