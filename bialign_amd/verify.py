"""Size-independent checks of engine output (host side, O(columns) per pair).

``rescore_trace`` walks a trace exactly like the reference's ``eval_affine_trace`` /
``eval_trace`` (bialignment.pyx:745-832) and returns the accumulated score, the
consumed lengths and the largest drift between the two alignments; for a
correct optimal trace the score equals ``optimize()``'s, the lengths are
(n, m, n, m) and the drift never exceeds ``max_shift``.
"""


def rescore_trace(codes, seq_a, cls_a, seq_b, cls_b, s1, s2, beta, gamma, delta, affine):
    idx = [0, 0, 0, 0]
    half = [2, 2]          # type of the last non-empty column per half: 0=(0,1) 1=(1,0) 2=(1,1)
    total = 0
    drift = 0
    for c in codes.tolist():
        y = ((c >> 3) & 1, (c >> 2) & 1, (c >> 1) & 1, c & 1)
        for t in range(4):
            idx[t] += y[t]
        mu = (int(s1[seq_a[idx[0] - 1], seq_b[idx[1] - 1]]) if y[0] and y[1] else 0,
              int(s2[cls_a[idx[2] - 1], cls_b[idx[3] - 1]]) if y[2] and y[3] else 0)
        shifts = abs(y[0] - y[2]) + abs(y[1] - y[3])
        if affine:
            score = delta * shifts
            for h in range(2):
                col = (y[2 * h], y[2 * h + 1])
                if col == (1, 1):
                    score += mu[h]
                    half[h] = 2
                elif col != (0, 0):
                    kind = 1 if col == (1, 0) else 0
                    score += gamma + (beta if half[h] != kind else 0)
                    half[h] = kind
        else:
            # non-affine column scores (pyx:233-248): a shift of any size costs Delta once
            score = (delta if shifts else 0)
            for h in range(2):
                col = (y[2 * h], y[2 * h + 1])
                score += mu[h] if col == (1, 1) else (gamma if col != (0, 0) else 0)
        total += score
        drift = max(drift, abs(idx[2] - idx[0]), abs(idx[3] - idx[1]))
    return total, tuple(idx), drift
