"""Predefined-graph helpers the STGCN predictor needs (reference model/STGCN/args.py:7-49, lib/predifineGraph.py:6-60): scaled graph
Laplacian and its Chebyshev polynomials, the adjacency of the PEMS distance csv, and a synthetic sensor graph for runs without the
(non-redistributable) data files."""
import csv

import numpy as np
import torch


def scaled_laplacian(W):
    """2 L / lambda_max - I with L = D - W normalised by sqrt(d_i d_j) where both degrees are positive (args.py:7-26)."""
    W = np.asarray(W, dtype=np.float64)
    n = W.shape[0]
    d = W.sum(axis=1)
    L = -W.copy()
    L[np.arange(n), np.arange(n)] = d
    pos = d > 0
    scale = np.ones((n, n))
    scale[np.ix_(pos, pos)] = np.sqrt(np.outer(d[pos], d[pos]))
    L = L / scale
    lam = np.linalg.eigvals(L).max().real
    return 2 * L / lam - np.identity(n)


def cheb_polynomials(L, Ks):
    """T_0 = I, T_1 = L, T_k = 2 L T_{k-1} - T_{k-2}  ->  (Ks, n, n)   (args.py:29-49)."""
    n = L.shape[0]
    if Ks < 1:
        raise ValueError("the spatial kernel size must be >= 1, got %r" % (Ks,))
    out = [np.identity(n)]
    if Ks > 1:
        out.append(np.array(L, dtype=np.float64))
    for _ in range(Ks - 2):
        out.append(2 * L @ out[-1] - out[-2])
    return np.stack(out, axis=0)


def adjacency_from_distance_csv(path, num_nodes):
    """Directed 0/1 adjacency from rows ``from,to,distance`` (header skipped), as lib/predifineGraph.py:46-60."""
    A = np.zeros((num_nodes, num_nodes), dtype=np.float32)
    with open(path) as f:
        f.readline()
        for row in csv.reader(f):
            if len(row) == 3:
                A[int(row[0]), int(row[1])] = 1
    return A


def synthetic_adjacency(num_nodes, seed=0, degree=3):
    """A connected sensor-like graph: a ring plus a few random chords (stand-in when the dataset's csv is absent)."""
    rng = np.random.RandomState(seed)
    A = np.zeros((num_nodes, num_nodes), dtype=np.float32)
    for i in range(num_nodes):
        A[i, (i + 1) % num_nodes] = 1
        for j in rng.choice(num_nodes, size=max(degree - 1, 0), replace=False):
            if j != i:
                A[i, j] = 1
    return A


def stgcn_graph(A, Ks=3):
    """The ``args_predictor.G`` tensor of the reference (args.py:86-88; the reference hard-codes 3 polynomials there)."""
    return torch.tensor(cheb_polynomials(scaled_laplacian(A), Ks), dtype=torch.float32)
