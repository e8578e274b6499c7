"""Lock-step multi-start solve: K bounded quasi-Newton searches advanced TOGETHER, one batched evaluation per tick.

The reference's entry point evaluates ONE plan per solver callback (src/mpc.py:202-255; its line search is switched off,
src/mpc.py:309), so a drop-in solve always runs the rollout at B = 1 -- where the GPU sits at a few per cent of its batched
rate -- and IPOPT "sometimes takes long to converge" from the single zero start (reference README.md:21).  Here K starts
(the zero start of src/mpc.py:292, the shifted previous plan, seeded samples of the input box) each run a projected L-BFGS
search with an Armijo backtracking line search; every start is a small state machine (trial point -> accept / shrink), and ONE
call of ``evaluate(X)`` per tick serves the trial points of all of them: a tick costs one B = K rollout instead of K sequential
B = 1 rollouts.  With ``torch.distributed`` initialised the K trial points are sharded over the ranks
(parallel.sharded_rollout): every rank runs this same deterministic host loop on the same gathered (cost, gradient) and
takes identical decisions -- this is the producer of the candidate batch SURVEY.md section 8e shards.

Not a re-implementation of Ipopt: optimiser results are unpinned (cyipopt 1.1.0 is neither vendored nor installed); what is
pinned is every value / gradient the search consumes.  Pure numpy on the host: the arrays are (K, H da).
"""
import numpy as np


def make_starts(n_starts, n, lb, ub, rng, warm=None, spread=1.0):
    """(K, n) start points: row 0 the reference's zero start (src/mpc.py:292-293), row 1 the previous plan shifted by one
    step when given, the rest uniform samples of the box (normal(0, spread) along unbounded directions)."""
    lb, ub = np.asarray(lb, dtype=np.float64), np.asarray(ub, dtype=np.float64)
    X = np.zeros((n_starts, n))
    k = 1
    if warm is not None and n_starts > 1:
        X[1] = warm
        k = 2
    if n_starts > k:
        finite = (lb > -1e15) & (ub < 1e15)
        lo, hi = np.where(finite, lb, -spread), np.where(finite, ub, spread)
        U = rng.uniform(lo, hi, size=(n_starts - k, n))
        G = rng.normal(0.0, spread, size=(n_starts - k, n))
        X[k:] = np.where(finite, U, G)
    return np.clip(X, lb, ub)


def _two_loop(g, S, Y, rho, cnt):
    """-H g for every start at once: L-BFGS two-loop recursion over the (K, m, n) histories, NEWEST pair at index 0 (plain
    slices, no gather); starts with fewer stored pairs skip the missing ones through `cnt`."""
    m = min(S.shape[1], int(cnt.max())) if len(cnt) else 0      # pairs stored by the start with the longest history
    q = g.copy()
    if m == 0:
        return -q
    alpha = []
    for j in range(m):                                      # newest pair first
        a = np.where(j < cnt, rho[:, j] * np.einsum("kn,kn->k", S[:, j], q), 0.0)
        alpha.append(a)
        q -= a[:, None] * Y[:, j]
    yy = np.einsum("kn,kn->k", Y[:, 0], Y[:, 0])
    ok = (cnt > 0) & (yy > 0)
    q *= np.where(ok, 1.0 / np.where(ok, rho[:, 0] * yy, 1.0), 1.0)[:, None]      # gamma = s.y / y.y of the newest pair
    for j in range(m - 1, -1, -1):                          # oldest pair first
        b = np.where(j < cnt, rho[:, j] * np.einsum("kn,kn->k", Y[:, j], q), 0.0)
        q += np.where(j < cnt, alpha[j] - b, 0.0)[:, None] * S[:, j]
    return -q


def lockstep_lbfgs(evaluate, X0, lb, ub, max_ticks=300, history=8, gtol=1e-4, ftol=1e-10, c1=1e-4, min_step=1e-12, patience=None,
                   line_points=1):
    """Minimise K copies of a bounded problem from the rows of X0, one ``evaluate`` per tick.

    evaluate(X (K, n)) -> (f (K,), g (K, n)); non-finite values are allowed (a plan whose risk-sensitive log-determinant
    does not exist returns NaN, src/mpc.py:183): such a trial point is rejected like a failed Armijo test, a start whose FIRST
    point is non-finite is dropped.  ``patience`` (ticks, optional): once the start that currently holds the lowest value has
    converged AND row 0 (the reference's own start) has finished, the others get this many more ticks before the search ends (a receding-horizon controller re-solves at the next
    step: the stragglers of a lock-step search are the starts least likely to matter).  ``line_points`` = S > 1: every tick evaluates S
    step lengths per start at once (alpha, alpha / 2, ... -- the line-search evaluations of a solver iteration as ONE batch of K S
    plans) and takes the lowest value that passes the Armijo test: fewer ticks, each a larger batch.  Returns (x_best, info) with info = {f (K,), x (K, n), ticks, evaluations, converged (K,),
    alive (K,), best}."""
    X = np.clip(np.asarray(X0, dtype=np.float64), lb, ub)
    K, n = X.shape
    lb = np.broadcast_to(np.asarray(lb, dtype=np.float64), (n,))
    ub = np.broadcast_to(np.asarray(ub, dtype=np.float64), (n,))
    F, G = evaluate(X)
    F, G = np.array(F, dtype=np.float64, copy=True).reshape(K), np.array(G, dtype=np.float64, copy=True).reshape(K, n)
    alive = np.isfinite(F) & np.isfinite(G).all(axis=1)
    F = np.where(alive, F, np.inf)
    G = np.where(alive[:, None], G, 0.0)
    m = history
    S, Y, rho = np.zeros((K, m, n)), np.zeros((K, m, n)), np.zeros((K, m))
    cnt = np.zeros(K, dtype=np.int64)
    done = ~alive
    iters = np.zeros(K, dtype=np.int64)

    def free_mask(x, g):                                     # components that may move: not pinned at a bound by the gradient
        return ~(((x <= lb) & (g > 0)) | ((x >= ub) & (g < 0)))

    def direction(x, g):
        fm = free_mask(x, g)
        gf = np.where(fm, g, 0.0)
        d = np.where(fm, _two_loop(gf, S, Y, rho, cnt), 0.0)
        slope = np.einsum("kn,kn->k", d, gf)
        bad = ~(slope < 0) | ~np.isfinite(d).all(axis=1)     # not a descent direction: steepest descent, history dropped
        if bad.any():
            d[bad] = -gf[bad]
            cnt[bad] = 0
        gn = np.sqrt(np.einsum("kn,kn->k", gf, gf))
        a0 = np.where(cnt == 0, np.minimum(1.0, 1.0 / np.where(gn > 0, gn, 1.0)), 1.0)
        return d, a0, np.abs(gf).max(axis=1)

    D, A, pg = direction(X, G)
    done |= pg <= gtol
    converged = done & alive
    XT = np.clip(X + A[:, None] * D, lb, ub)
    ticks, best_done_at = 0, None
    while ticks < max_ticks and not done.all():
        if patience is not None:
            if converged[int(np.argmin(F))] and done[0]:          # (row 0, the reference's own zero start, always runs to its end)
                best_done_at = ticks if best_done_at is None else best_done_at
                if ticks - best_done_at >= patience:
                    break
            else:
                best_done_at = None
        S_ = int(line_points)
        if S_ <= 1:
            ft, gt = evaluate(np.where(done[:, None], X, XT))
            ft, gt = np.asarray(ft, dtype=np.float64).reshape(K), np.asarray(gt, dtype=np.float64).reshape(K, n)
            step = XT - X
            ok = np.isfinite(ft) & np.isfinite(gt).all(axis=1) & ~done
            ok &= ft <= F + c1 * np.einsum("kn,kn->k", G, step)
        else:
            # S step lengths per start in one batch of K S plans, row s K + k = start k at alpha_k / 2^s
            scale = 0.5 ** np.arange(S_)
            XS = np.clip(X[None] + (A[None, :, None] * scale[:, None, None]) * D[None], lb, ub)
            XS = np.where(done[None, :, None], X[None], XS)
            fs, gs = evaluate(XS.reshape(S_ * K, n))
            fs, gs = np.asarray(fs, dtype=np.float64).reshape(S_, K), np.asarray(gs, dtype=np.float64).reshape(S_, K, n)
            steps = XS - X[None]
            good = np.isfinite(fs) & np.isfinite(gs).all(axis=2) & (fs <= F[None] + c1 * np.einsum("kn,skn->sk", G, steps))
            pick = np.argmin(np.where(good, fs, np.inf), axis=0)          # the lowest value among the step lengths that pass
            rows = np.arange(K)
            ok = good[pick, rows] & ~done
            ft, gt, XT = fs[pick, rows], gs[pick, rows], XS[pick, rows]
            step = XT - X
            A = np.where(ok, A * scale[pick], A * scale[-1])              # (a failing start continues below its smallest trial)
        ticks += 1
        shrink = ~ok & ~done
        if ok.any():
            i = np.where(ok)[0]
            s, y = step[i], gt[i] - G[i]
            sy = np.einsum("kn,kn->k", s, y)
            good = sy > 1e-10 * np.sqrt(np.einsum("kn,kn->k", s, s) * np.einsum("kn,kn->k", y, y))
            ig = i[good]
            S[ig, 1:], Y[ig, 1:], rho[ig, 1:] = S[ig, :-1], Y[ig, :-1], rho[ig, :-1]       # newest pair at index 0
            S[ig, 0], Y[ig, 0], rho[ig, 0] = s[good], y[good], 1.0 / sy[good]
            cnt[ig] = np.minimum(cnt[ig] + 1, m)
            small = (F[i] - ft[i]) <= ftol * np.maximum(np.maximum(np.abs(F[i]), np.abs(ft[i])), 1.0)
            X[i], F[i], G[i] = XT[i], ft[i], gt[i]
            iters[i] += 1
            Dn, An, pgn = direction(X, G)
            D[i], A[i] = Dn[i], An[i]
            fin = np.zeros(K, dtype=bool)
            fin[i] = small | (pgn[i] <= gtol)
            converged |= fin
            done |= fin
        if shrink.any():
            A[shrink] *= 0.5
            stalled = shrink & (A * np.abs(D).max(axis=1) < min_step)
            converged |= stalled                              # no representable descent step left: a stationary point to rounding
            done |= stalled
        XT = np.clip(X + A[:, None] * D, lb, ub)
    best = int(np.argmin(F)) if np.isfinite(F).any() else 0
    info = {"f": F, "x": X, "ticks": ticks, "evaluations": ticks + 1, "converged": converged, "alive": alive, "best": best,
            "iterations": iters}
    return X[best].copy(), info
