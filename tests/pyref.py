"""Second, independent restatement of the reference's per-read algorithm in plain Python (small cases only).

Test infrastructure.  It exists to (a) author the hand-derived golden vectors under tests/golden/ as *code that
spells out every float32 operation*, and (b) cross-check the C oracle (oracle/rappas_oracle.c) with an implementation
that shares no code with it.  Line references are to the reference sources (paths relative to its root):
src/core/algos/PlacementProcess.java, src/core/algos/AmbigSequenceKnife.java, src/core/DNAStatesShifted.java,
src/core/AAStates.java, src/main_v2/Main_DBBUILD_3.java.
"""
import math

import numpy as np

f32 = np.float32

DNA_STATES = {"A": 0, "T": 1, "U": 1, "C": 2, "G": 3}                       # DNAStatesShifted.java:182-209
DNA_AMBIG = {                                                               # DNAStatesShifted.java:62-96
    "R": "AG", "Y": "CT", "S": "CG", "W": "AT", "K": "GT", "M": "AC",
    "B": "CGT", "D": "AGT", "H": "ACT", "V": "ACG", "N": "ACGT",
}
AA_ORDER = "RHKDESTNQCGPAILMFWYV"                                           # AAStates.java:23-28
AA_AMBIG = {"B": "DN", "Z": "EQ", "J": "IL"}                                # AAStates.java:104-109


def thresholds(omega, n_states, k):
    """Main_DBBUILD_3.java:165-166 (omega is a float: ArgumentsParser_v2.java:52)."""
    ratio = f32(omega) / f32(n_states)
    p = f32(math.pow(0.0 + float(ratio), k))
    return p, f32(math.log10(float(p)))


def classify(alphabet, ch, convert_uo=False):
    """-> ('state', s) | ('amb', [alternatives]) | ('bad', None)   (isAmbiguous is tested first: AmbigSequenceKnife.java:106)"""
    if alphabet == 4:
        up = ch.upper()
        if up in DNA_AMBIG:
            return "amb", [DNA_STATES[c] for c in DNA_AMBIG[up]]
        if ch in ".-":
            return "amb", [0, 0, 0, 0]                                      # byte[4] never filled (:57-58)
        if up in DNA_STATES:
            return "state", DNA_STATES[up]
        return "bad", None
    if ch in "-*!Xx":
        return "amb", list(range(20))                                       # AAStates.java:97-103
    up = ch.upper()
    if up in AA_AMBIG:
        return "amb", [AA_ORDER.index(c) for c in AA_AMBIG[up]]
    if up in AA_ORDER:
        return "state", AA_ORDER.index(up)
    if convert_uo and up in "UO":                                           # AAStates.java:118-123
        return "state", 9 if up == "U" else 14
    return "bad", None


def compress_mer_dna(states):
    """DNAStatesShifted.java:115-143: base i -> bits 2*(i%4) of byte i//4."""
    out = [0] * ((len(states) + 3) // 4)
    for i, s in enumerate(states):
        out[i // 4] |= (s << (2 * (i % 4))) & 0xFF
    return out


def kmer_code(alphabet, states):
    if alphabet == 4:
        return int.from_bytes(bytes(compress_mer_dna(states)), "little")
    return sum(s << (5 * i) for i, s in enumerate(states))


class JavaPriorityQueue:
    """java.util.PriorityQueue (Comparable path) over (node, score) with Score.compareTo = Float.compare."""

    def __init__(self):
        self.q = []

    @staticmethod
    def cmp(a, b):
        a, b = f32(a), f32(b)
        if a < b:
            return -1
        if a > b:
            return 1
        ia, ib = int(a.view(np.int32)), int(b.view(np.int32))
        return 0 if ia == ib else (-1 if ia < ib else 1)

    def add(self, e):
        k = len(self.q)
        self.q.append(e)
        while k > 0:                                                        # siftUp
            parent = (k - 1) >> 1
            if self.cmp(e[1], self.q[parent][1]) >= 0:
                break
            self.q[k] = self.q[parent]
            k = parent
        self.q[k] = e

    def poll(self):
        res = self.q[0]
        x = self.q.pop()
        n = len(self.q)
        if n > 0:                                                           # siftDown
            k, half = 0, n >> 1
            while k < half:
                child = 2 * k + 1
                c = self.q[child]
                right = child + 1
                if right < n and self.cmp(c[1], self.q[right][1]) > 0:
                    child = right
                    c = self.q[child]
                if self.cmp(x[1], c[1]) <= 0:
                    break
                self.q[k] = c
                k = child
            self.q[k] = x
        return res


def place_read(db, read, keep_at_most=7, keep_factor=0.01, amb_mode="mean", ns_bound=float("-inf"), convert_uo=False):
    """db: dict(alphabet, k, n_branches, T (f32), P (f32), rows {code: [(branch, f32 score), ...]}).
    Returns dict(rows=[(branch, score f32, lwr float)], flags=set(...), S={branch: f32}, L=[...], H=int)."""
    alphabet, k = db["alphabet"], db["k"]
    T, P = f32(db["T"]), f32(db["P"])
    flags = set()
    R = len(read)
    seq, alts = [], {}
    amb = [0] * R
    for i, ch in enumerate(read):                                           # AmbigSequenceKnife.java:103-130
        kind, val = classify(alphabet, ch, convert_uo)
        if kind == "bad":
            flags.add("bad_char")
            seq.append(0)
        elif kind == "amb":
            flags.add("ambiguous")
            for j in range(i - k + 1, i + 1):
                if -1 < j < R:
                    amb[j] += 1
            seq.append(-1)
            alts[i] = val
        else:
            seq.append(val)
    if R < k:
        flags.add("too_short")
    if "bad_char" in flags or R < k:
        return dict(rows=[], flags=flags, S={}, L=[], H=0)
    Q = R - k + 1
    max_amb = int(math.floor(math.pow(k, 1.0 / alphabet)))                  # AmbigSequenceKnife.java:95
    S, C, L, H = {}, {}, [], 0
    QT = f32(Q) * T                                                         # int*float (PlacementProcess.java:728)

    def touch(x):
        if C.get(x, 0) == 0:
            L.append(x)
            S[x] = f32(f32(S.get(x, f32(0.0))) + QT)
        C[x] = C.get(x, 0) + 1

    for j in range(Q):
        if amb[j] < 1:
            row = db["rows"].get(kmer_code(alphabet, seq[j:j + k]))
            if row is None:
                continue
            for x, v in row:                                                # :719-735
                touch(x)
                S[x] = f32(S[x] + f32(f32(v) - T))
                H += 1
        elif amb[j] > max_amb or amb_mode == "skip":
            continue
        else:
            # AmbigSequenceKnife.java:235-260: word number `jump` takes alt_i[jump % len_i] at EVERY ambiguous position i
            # (the counter restarts for each position), altProduct words in all -- not a cartesian product unless the counts
            # are coprime; one ambiguous position (all that DNA k < 16 and proteins allow) gives its alternatives in order
            ps = [i for i in range(k) if seq[j + i] == -1]
            W = 1
            for p in ps:
                W *= len(alts[j + p])
            S_amb, C_amb, L_amb = {}, {}, []
            for jump in range(W):
                w = list(seq[j:j + k])
                for p in ps:
                    w[p] = alts[j + p][jump % len(alts[j + p])]
                row = db["rows"].get(kmer_code(alphabet, w))
                if row is None:
                    continue
                for x, v in row:
                    H += 1
                    if C_amb.get(x, 0) == 0:
                        L_amb.append(x)
                        if amb_mode == "max":
                            S_amb[x] = f32(v)
                    C_amb[x] = C_amb.get(x, 0) + 1
                    if amb_mode == "mean":                                  # :1155 float += double
                        S_amb[x] = f32(float(S_amb.get(x, f32(0.0))) + math.pow(10.0, float(f32(v))))
                    elif f32(v) > S_amb[x]:                                 # :1215
                        S_amb[x] = f32(v)
            for x in L_amb:
                if C.get(x, 0) == 0:
                    L.append(x)
                    S[x] = QT                                               # :1165 / :1226 (assignment)
                C[x] = C.get(x, 0) + 1
                if amb_mode == "mean":
                    avg = f32(f32(S_amb[x] + f32(f32(W - C_amb[x]) * P)) / f32(W))            # :1168
                    S[x] = f32(float(S[x]) + (math.log10(float(avg)) - float(T)))             # :1169
                else:
                    S[x] = f32(S[x] + f32(S_amb[x] - T))                    # :1230
    if not L:
        return dict(rows=[], flags=flags, S=S, L=L, H=H)
    flags.add("placed")
    K = keep_at_most
    num_best = min(K, len(L))
    pq = JavaPriorityQueue()                                                # fillBestScoreList :396-451
    for x in L:
        pq.add((x, S[x]))
        if len(pq.q) > num_best:
            pq.poll()
    total = 0.0
    lowest, best = f32(0.0), f32(-3.4028234663852886e38)
    for x, s in pq.q:
        total += math.pow(10.0, float(s))
        if s < lowest:
            lowest = s
        if s > best:
            best = s
    best_list = list(pq.q) + [(-1, f32(-np.inf))] * (K - len(pq.q))
    # Arrays.sort: stable ascending by Float.compare
    best_list = sorted(best_list, key=lambda e: (float(e[1]), 0 if not (e[1] == 0 and np.signbit(e[1])) else -1))
    shift = best if f32(-308.0) >= lowest else f32(0.0)
    if shift != 0:
        total = 0.0
        for ii in range(K - num_best, K):
            total += math.pow(10.0, float(f32(best_list[ii][1] - shift)))   # float subtract then widen (:445-447)
    if not (best_list[K - 1][1] >= f32(ns_bound)):                          # :974
        flags.add("below_nsbound")
        return dict(rows=[], flags=flags, S=S, L=L, H=H)
    best2, lowest2 = best_list[K - 1][1], best_list[K - num_best][1]
    shift2 = best2 if f32(-308.0) >= lowest2 else f32(0.0)                  # :978-980
    rows, best_ratio = [], -1.0
    for i in range(K - 1, K - num_best - 1, -1):                            # :984-1025
        ratio = math.pow(10.0, float(best_list[i][1]) - float(shift2)) / total
        if i == K - 1:
            best_ratio = ratio
        if i < K - 1 and ratio < best_ratio * float(f32(keep_factor)):
            break
        rows.append((best_list[i][0], best_list[i][1], ratio))
    return dict(rows=rows, flags=flags, S=S, L=L, H=H)


def db_to_csr(db):
    """dict-of-rows toy DB -> the flat arrays of the C ABI (rows in insertion order of the dict)."""
    codes = list(db["rows"].keys())
    off, br, sc = [0], [], []
    for c in codes:
        for x, v in db["rows"][c]:
            br.append(x)
            sc.append(v)
        off.append(len(br))
    return (np.array(codes, np.uint64), np.array(off, np.uint64), np.array(br, np.uint16), np.array(sc, np.float32))
