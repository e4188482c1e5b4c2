"""Queue-order statistics of LSD region growing (analysis only; reads the oracle's growth log).

For every region_grow call the oracle logs the queue (x, y, index of the entry that added the pixel).  The script replays the queue
under a few "how many entries can one round of k_lsd_grow* serve" models and prints the number of rounds each needs.
"""
import ctypes as C
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
sys.path.insert(0, os.path.dirname(__file__))
import oracle_lib as ol
import synth_frames as sf


def growlog(img):
    L = ol.load()
    cap = 1 << 27
    out = np.empty(cap, np.int32)
    L.pso_lsd_growlog.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    n = L.pso_lsd_growlog(img.ctypes.data, img.shape[1], img.shape[0], img.strides[0], out.ctypes.data, cap)
    assert n <= cap
    g = out[:n]
    regs = []
    p = 0
    while p < n:
        if g[p] == -2:
            print('  candidate tests (defined, not used):', g[p + 1])
            break
        assert g[p] == -1
        m = g[p + 1]
        regs.append(g[p + 2:p + 2 + 3 * m].reshape(m, 3))
        p += 2 + 3 * m
    return regs


def rounds_batch(reg, cap=7):
    # current scheme: a round pops min(cap, entries present at its start)
    n = len(reg); par = reg[:, 2]
    i = 0; size = 1; r = 0
    # size after processing entries < j = 1 + #(par < j)
    cnt = np.bincount(par[1:], minlength=n) if n > 1 else np.zeros(n, int)
    csum = np.concatenate([[1], 1 + np.cumsum(cnt)])  # csum[j] = size after entries < j processed
    while i < n:
        size = csum[i]
        nb = min(cap, size - i)
        i += nb; r += 1
    return r


def rounds_window(reg, wx=8, wy=8, place='centre'):
    # a round loads a wx x wy window; entries are popped in order while the entry and its 3x3 are inside
    n = len(reg); i = 0; r = 0
    x = reg[:, 0]; y = reg[:, 1]
    sx, sy = x[0], y[0]
    while i < n:
        ex, ey = x[i], y[i]
        if place == 'centre':
            ox, oy = ex - wx // 2 + 1, ey - wy // 2 + 1
        elif place == 'down':      # seeds are the top of their region: look ahead downwards
            ox, oy = ex - wx // 2 + 1, ey - 1
        elif place == 'away':      # away from the seed
            dx, dy = ex - sx, ey - sy
            ox = ex - 1 if dx > abs(dy) else (ex - wx + 2 if -dx > abs(dy) else ex - wx // 2 + 1)
            oy = ey - 1 if dy >= abs(dx) else (ey - wy + 2 if -dy > abs(dx) else ey - wy // 2 + 1)
            if i == 0: ox, oy = ex - wx // 2 + 1, ey - 1
        elif place == 'corner':    # per axis: ahead of the growth direction (seed -> entry)
            dx, dy = ex - sx, ey - sy
            m = max(abs(dx), abs(dy), 1)
            ox = ex - 1 if dx > 0.5 * m else (ex - wx + 2 if -dx > 0.5 * m else ex - wx // 2 + 1)
            oy = ey - 1 if dy > 0.5 * m else (ey - wy + 2 if -dy > 0.5 * m else ey - wy // 2 + 1)
            if i == 0: ox, oy = ex - wx // 2 + 1, ey - 1
        r += 1
        while i < n and ox + 1 <= x[i] <= ox + wx - 2 and oy + 1 <= y[i] <= oy + wy - 2:
            i += 1
    return r


if __name__ == '__main__':
    kinds = sys.argv[1:] or ['struct', 'desk']
    for kind in kinds:
        gray = np.ascontiguousarray(sf.Scene(640, 480, kind, 3).gray(2))
        regs = growlog(gray)
        sizes = np.array([len(r) for r in regs])
        print(kind, 'regions', len(regs), 'pixels', sizes.sum(), 'singles', (sizes == 1).sum(), '>=15:', (sizes >= 15).sum())
        print('  rounds current(7)', sum(rounds_batch(r) for r in regs))
        for (wx, wy) in [(8, 8)]:
            for place in ['down', 'away', 'corner']:
                print('  window %2dx%-2d %-6s' % (wx, wy, place), sum(rounds_window(r, wx, wy, place) for r in regs))
