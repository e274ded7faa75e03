"""icp_amd/newton.py, the non-interactive twin of the reference's Gauss-Newton toy (newton.py:30-78).
The reference checks its analytic Jacobian against central differences in commented-out lines
(newton.py:44-51); that check is live here.  From the reference's own initial guess (2,1,8,1) the
undamped step does not converge (it wanders off after a few steps, there as here); close to the
target it converges quadratically."""
import numpy as np

from icp_amd import newton


def test_jacobian_is_the_derivative():
    x = np.linspace(0, newton.data_range, newton.data_len)
    p = np.array(newton.GUESS)
    J = newton.Jacobian(x, *p)
    h = 1e-4                                            # newton.py:15
    for k in range(4):
        e = np.zeros(4)
        e[k] = h
        fd = (newton.Estimate(x, *(p + e)) - newton.Estimate(x, *(p - e))) / (2 * h)
        np.testing.assert_allclose(J[:, k], fd, rtol=1e-6, atol=1e-6)


def test_first_step_matches_normal_equations():
    x = np.linspace(0, newton.data_range, newton.data_len)
    target = newton.GenerateTarget(x)
    delta, err = newton.Step(x, target, np.array(newton.GUESS))
    r = target - newton.Estimate(x, *newton.GUESS)
    ref = np.linalg.lstsq(newton.Jacobian(x, *newton.GUESS), r, rcond=None)[0]
    np.testing.assert_allclose(delta, ref, rtol=1e-9)
    np.testing.assert_allclose(err, (r * r).sum())


def test_converges_near_the_target():
    p, hist = newton.GaussNewton(p0=(4.6, 1.05, 9.5, 1.97), steps=12)
    np.testing.assert_allclose(p, newton.TARGET, rtol=1e-10)
    errs = [e for _, e, _ in hist]
    assert errs[-1] < 1e-20 and errs[3] < 1e-3 * errs[0]
