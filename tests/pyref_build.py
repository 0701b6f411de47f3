"""Independent plain-Python restatement of the reference's phylo-kmer construction loop (small cases only).

Test infrastructure: cross-checks oracle/rappas_build_oracle.c with code that shares nothing with it.  Follows
src/core/algos/WordExplorer_v3.java:98-199 (recursion and its float32 running sum), src/main_v2/Main_DBBUILD_3.java:693-712
(one explorer per (node, pos), every first state in turn) and src/core/hash/CustomHash_v4_FastUtil81.java:73-89 (max per
(word, branch))."""
import sys

import numpy as np

f32 = np.float32


class WordExplorer:
    def __init__(self, tab, node, branch, k, T, bits, gaps, limit1, sink):
        self.states, self.pp = tab
        self.node, self.branch, self.k, self.T, self.bits = node, branch, k, f32(T), bits
        self.gaps, self.limit1, self.sink = gaps, limit1, sink
        self.sum = f32(0.0)
        self.word = [0] * k
        self.bound, self.bound_k, self.cur_k, self.first_jump = False, -1, 0, -1
        self.visits = 0

    def explore(self, i, j):
        n_sites, n_states = self.pp.shape[1], self.pp.shape[2]
        if i > n_sites - 1:
            return
        if self.cur_k == 0:
            self.first_jump = -1
        self.visits += 1
        self.word[self.cur_k] = int(self.states[self.node, i, j])
        p = float(self.pp[self.node, i, j])                       # getPP returns a double
        self.sum = f32(float(self.sum) + p)                       # float += double
        self.bound = bool(self.sum < self.T)
        if self.bound:
            self.bound_k = self.cur_k
        if self.cur_k == self.k - 1:
            if not self.bound:
                code = sum(s << (self.bits * t) for t, s in enumerate(self.word))
                key = (code, self.branch)
                old = self.sink.get(key)
                if old is None or self.sum > old:
                    self.sink[key] = self.sum
                self.sink["#"] = self.sink.get("#", 0) + 1
            self.sum = f32(float(self.sum) - p)
            return
        for j2 in range(n_states):
            if self.bound and self.bound_k == self.cur_k + 1:
                break
            self.cur_k += 1
            self.explore(i + 1, j2)
            self.cur_k -= 1
            if self.gaps is not None and i < n_sites - 1 and self.gaps[i + 1]:
                if not self.limit1:
                    for g in self.gaps[i + 1]:
                        self.cur_k += 1
                        self.explore(i + 1 + g, j2)
                        self.cur_k -= 1
                elif self.first_jump == -1:
                    self.first_jump = i
                    for g in self.gaps[i + 1]:
                        self.cur_k += 1
                        self.explore(i + 1 + g, j2)
                        self.cur_k -= 1
        self.sum = f32(float(self.sum) - p)


def build(alphabet, k, states, pp, node_branch, T, gaps=None, limit1=True):
    """-> (dict {(code, branch): f32 score}, tuples, visits); gaps = list (per site) of lists of interval lengths, or None."""
    sys.setrecursionlimit(10000)
    bits = 2 if alphabet == 4 else 5
    sink, visits = {}, 0
    n_nodes, n_sites, n_states = pp.shape
    for node in range(n_nodes):
        for pos in range(n_sites - k + 2):
            we = WordExplorer((states, pp), node, int(node_branch[node]), k, T, bits, gaps, limit1, sink)
            for j in range(n_states):
                we.explore(pos, j)
            visits += we.visits
    tuples = sink.pop("#", 0)
    return sink, tuples, visits


def to_csr(sink):
    keys = sorted(sink)
    codes = sorted({c for c, _ in keys})
    off, br, sc = [0], [], []
    it = iter(keys)
    by = {}
    for c, b in keys:
        by.setdefault(c, []).append(b)
    for c in codes:
        for b in by[c]:
            br.append(b)
            sc.append(sink[(c, b)])
        off.append(len(br))
    return (np.array(codes, np.uint64), np.array(off, np.uint64), np.array(br, np.uint16), np.array(sc, np.float32))
