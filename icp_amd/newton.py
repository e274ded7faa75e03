"""Non-interactive twin of the reference's Gauss-Newton toy (newton.py:30-78): fit
a*sin(b*x) + c*cos(d*x) to samples of 5*sin(x) + 10*cos(2x) by plain Gauss-Newton steps
delta = (J^T J)^-1 J^T r.  Four parameters, twenty samples: host arithmetic (numpy), nothing
here touches the GPU -- it is kept as the smallest end-to-end check of the normal-equation step
that icp_amd's LM-CG solver (nicp.hip) performs at scale.
The reference prompts "Continue ?" after every step; `GaussNewton` takes a step count."""
import numpy as np

# newton.py:13-25
data_range = 5
data_len = 20
TARGET = (5.0, 1.0, 10.0, 2.0)
GUESS = (2.0, 1.0, 8.0, 1.0)


def GenerateTarget(x, params=TARGET):
    """newton.py:30-32"""
    A, B, C, D = params
    return A * np.sin(B * x) + C * np.cos(D * x)


def Estimate(x, a, b, c, d):
    """newton.py:34-36"""
    return a * np.sin(b * x) + c * np.cos(d * x)


def Jacobian(x, a, b, c, d):
    """newton.py:38-52: d(Estimate)/d(a,b,c,d), one row per sample."""
    J = np.zeros([x.shape[0], 4])
    J[:, 0] = np.sin(b * x)
    J[:, 1] = a * x * np.cos(b * x)
    J[:, 2] = np.cos(d * x)
    J[:, 3] = -c * x * np.sin(d * x)
    return J


def Step(x, target, p):
    """One pass of the loop body newton.py:61-68: (delta, sum of squared residuals at p)."""
    r = target - Estimate(x, *p)
    J = Jacobian(x, *p)
    delta = np.linalg.inv(J.T @ J) @ J.T @ r
    return delta, float((r * r).sum())


def GaussNewton(p0=GUESS, steps=20, x=None, target=None, verbose=False):
    """The reference's main loop (newton.py:57-78) for a fixed number of steps.
    Returns (parameters, history of (parameters, error, delta))."""
    x = np.linspace(0, data_range, data_len) if x is None else np.asarray(x, dtype=np.float64)
    target = GenerateTarget(x) if target is None else np.asarray(target, dtype=np.float64)
    p = np.array(p0, dtype=np.float64)
    hist = []
    for _ in range(steps):
        delta, err = Step(x, target, p)
        if verbose:
            print("[", p[0], p[1], p[2], p[3], "]", err, delta)     # newton.py:69
        hist.append((p.copy(), err, delta))
        p = p + delta
    return p, hist


if __name__ == "__main__":
    GaussNewton(verbose=True)
