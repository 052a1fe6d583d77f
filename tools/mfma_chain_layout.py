#!/usr/bin/env python
"""Layout check for the round-2 encoder design: with the operand layout of v_mfma_f32_16x16x32_bf16 that
tools/mfma_layout_probe.hip verified on hardware,
    A[m = l & 15][k = 8 (l >> 4) + j],   B[k = 8 (l >> 4) + j][n = l & 15],   D[m = 4 (l >> 4) + r][n = l & 15],
the accumulator registers of layer i can be fed to layer i+1 as its B fragments WITHOUT leaving the lane - no LDS round trip and
no shuffle - if layer i+1's packed weights use the K order the accumulators come in: the 8 B values of lane l for K-step ks are
the 4 + 4 accumulator values of feature tiles 2 ks and 2 ks + 1, i.e. feature phi(ks, g, j) = (2 ks + j // 4) * 16 + 4 g + j % 4.
This script emulates the instruction lane by lane in numpy and checks a three-layer chain against plain matmuls."""
import numpy as np


def mfma(a, b, c):
    """a, b: [64 lanes, 8], c: [64 lanes, 4] -> D in the accumulator layout (fp32 accumulate)"""
    lanes = np.arange(64)
    A = np.zeros((16, 32)); B = np.zeros((32, 16))
    for l in lanes:
        A[l & 15, 8 * (l >> 4):8 * (l >> 4) + 8] = a[l]
        B[8 * (l >> 4):8 * (l >> 4) + 8, l & 15] = b[l]
    D = A @ B
    out = c.copy()
    for l in lanes:
        for r in range(4):
            out[l, r] += D[4 * (l >> 4) + r, l & 15]
    return out


def phi(ks, g, j):
    return (2 * ks + j // 4) * 16 + 4 * g + j % 4


def pack_chained(W):
    """[M, K] weights of a layer whose input arrives as accumulator registers: fragment (mt, ks, lane)[j] = W[mt*16 + (lane & 15)][phi(ks, lane >> 4, j)]"""
    M, K = W.shape
    out = np.zeros((M // 16, K // 32, 64, 8))
    for mt in range(M // 16):
        for ks in range(K // 32):
            for l in range(64):
                for j in range(8):
                    out[mt, ks, l, j] = W[mt * 16 + (l & 15), phi(ks, l >> 4, j)]
    return out


def pack_plain(W):
    """the packing of include/quadswarm_encoder.h (input from LDS rows, K contiguous)"""
    M, K = W.shape
    out = np.zeros((M // 16, K // 32, 64, 8))
    for mt in range(M // 16):
        for ks in range(K // 32):
            for l in range(64):
                out[mt, ks, l] = W[mt * 16 + (l & 15), ks * 32 + 8 * (l >> 4):ks * 32 + 8 * (l >> 4) + 8]
    return out


def layer_from_rows(Wp, X):
    """X: [16 rows, K] activations as LDS rows -> accumulators acc[mt][64 lanes, 4] of all M/16 feature tiles"""
    MT, KS = Wp.shape[:2]
    acc = [np.zeros((64, 4)) for _ in range(MT)]
    for ks in range(KS):
        b = np.zeros((64, 8))
        for l in range(64):
            b[l] = X[l & 15, ks * 32 + 8 * (l >> 4):ks * 32 + 8 * (l >> 4) + 8]
        for mt in range(MT):
            acc[mt] = mfma(Wp[mt, ks], b, acc[mt])
    return acc


def layer_from_acc(Wp, prev):
    """prev: accumulators of the layer before (after its activation), used in place as B fragments"""
    MT, KS = Wp.shape[:2]
    assert len(prev) == 2 * KS
    acc = [np.zeros((64, 4)) for _ in range(MT)]
    for ks in range(KS):
        b = np.concatenate([prev[2 * ks], prev[2 * ks + 1]], axis=1)      # the lane's own 4 + 4 values: no data movement
        for mt in range(MT):
            acc[mt] = mfma(Wp[mt, ks], b, acc[mt])
    return acc


def acc_to_matrix(acc):
    """accumulators -> [features, 16 rows]"""
    out = np.zeros((16 * len(acc), 16))
    for mt, a in enumerate(acc):
        for l in range(64):
            for r in range(4):
                out[mt * 16 + 4 * (l >> 4) + r, l & 15] = a[l, r]
    return out


if __name__ == "__main__":
    rng = np.random.RandomState(0)
    X = rng.randn(16, 32)
    W1, W2, W3 = rng.randn(64, 32) * 0.3, rng.randn(96, 64) * 0.3, rng.randn(32, 96) * 0.3
    h1 = layer_from_rows(pack_plain(W1), X)
    h1 = [np.tanh(a) for a in h1]
    h2 = [np.tanh(a) for a in layer_from_acc(pack_chained(W2), h1)]
    h3 = layer_from_acc(pack_chained(W3), h2)
    want = W3 @ np.tanh(W2 @ np.tanh(W1 @ X.T))
    err = np.abs(acc_to_matrix(h3) - want).max()
    print("three chained layers, accumulators used as B fragments in place: max abs err vs matmul =", err)
    assert err < 1e-12
    ks, seen = np.arange(2), set()
    for k in range(3):
        for g in range(4):
            for j in range(8):
                seen.add(phi(k, g, j))
    assert seen == set(range(96))            # phi is a bijection onto the layer's input features
    print("phi covers every input feature exactly once")
