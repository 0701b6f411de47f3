"""Parity checker: HIP engine output vs the CPU oracle on the same inputs."""
import numpy as np

from oracle import oracle as O

TIE = O.RO_FLAG_TIE
LWR_RTOL = 1e-9  # north_star asks 1e-5 relative; double pow/log differences are ~1e-16


def compare_with_oracle(got, ref, odb=None, seq=None, off=None, amb_mode=O.AMB_MEAN, max_report=5):
    """got: Placements (engine); ref: dict from OracleDB.place.  Returns a dict of stats; raises AssertionError
    with the first mismatching reads otherwise.

    Bar: flags equal (TIE is oracle-only), n_rows equal, scores bit-equal, branches equal, LWR within LWR_RTOL.
    Reads the oracle flags as TIE (exact float tie inside the top-(K+1)): the reference's order among equal scores
    depends on its hash-map layout, the engine uses (score desc, branch asc); for those reads the score/LWR rows
    must still be bit-equal / within tolerance and every reported (branch, score) must be a true (branch, S[branch])
    pair of the oracle's score vector."""
    n = len(ref["n_rows"])
    rflags = ref["flags"] & ~np.uint32(TIE)
    bad = []
    gflags = got.flags.astype(np.uint32)
    fl_bad = np.nonzero(gflags != rflags)[0]
    for r in fl_bad[:max_report]:
        bad.append(f"read {r}: flags got {gflags[r]:#x} want {rflags[r]:#x}")
    nr_bad = np.nonzero(got.n_rows != ref["n_rows"])[0]
    for r in nr_bad[:max_report]:
        bad.append(f"read {r}: n_rows got {got.n_rows[r]} want {ref['n_rows'][r]} scores got {got.score[r]} want {ref['score'][r]} "
                   f"lwr got {got.lwr[r]} want {ref['lwr'][r]}")
    sc_bad = np.nonzero((got.score.view(np.uint32) != ref["score"].view(np.uint32)).any(axis=1))[0]
    for r in sc_bad[:max_report]:
        bad.append(f"read {r}: score bits differ got {got.score[r]} want {ref['score'][r]} (branches {got.branch[r]} vs {ref['branch'][r]})")
    with np.errstate(invalid="ignore", divide="ignore"):
        denom = np.maximum(np.abs(ref["lwr"]), 1e-300)
        rel = np.abs(got.lwr - ref["lwr"]) / denom
    rel[ref["lwr"] == got.lwr] = 0
    lw_bad = np.nonzero((rel > LWR_RTOL).any(axis=1))[0]
    for r in lw_bad[:max_report]:
        bad.append(f"read {r}: lwr got {got.lwr[r]} want {ref['lwr'][r]}")
    br_diff = np.nonzero((got.branch != ref["branch"]).any(axis=1))[0]
    tie = (ref["flags"] & TIE) != 0
    ties_resolved = 0
    for r in br_diff:
        if not tie[r]:
            bad.append(f"read {r}: branches got {got.branch[r]} want {ref['branch'][r]} scores {ref['score'][r]}")
            continue
        if odb is None:
            continue
        S, _, _ = odb.score_vector(bytes(seq[int(off[r]):int(off[r + 1])]), amb_mode)
        for i in range(int(got.n_rows[r])):
            b = int(got.branch[r, i])
            if not (b < len(S) and np.float32(S[b]).view(np.uint32) == got.score[r, i].view(np.uint32)):
                bad.append(f"tie read {r}: row {i} reports branch {b} score {got.score[r, i]} but oracle S[b]={S[b] if b < len(S) else None}")
        if len(set(got.branch[r, :int(got.n_rows[r])].tolist())) != int(got.n_rows[r]):
            bad.append(f"tie read {r}: duplicate branches {got.branch[r]}")
        ties_resolved += 1
    assert not bad, f"{len(bad)} mismatch lines (of {n} reads):\n" + "\n".join(bad[:4 * max_report])
    return dict(n=n, ties=int(tie.sum()), tie_branch_diffs=ties_resolved, max_lwr_rel=float(rel.max()) if n else 0.0,
                placed=int((rflags & 1).sum()))
