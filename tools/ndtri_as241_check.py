#!/usr/bin/env python
"""Accuracy of the two AS 241 branches used by csrc/problem.h:ndtri_as241_core against
scipy.special.ndtri (the host function of PRIOR_NORMAL), in NumPy double arithmetic.
The far tail (sqrt(-ln p) > 5, p < 1.4e-11) is not restated: the device takes ocml erfcinv there."""
import numpy as np
from scipy.special import ndtri


def flog(x):
    """csrc/problem.h:log_pos in NumPy double arithmetic."""
    m, e = np.frexp(x)
    lo = m < 0.7071067811865476
    m = np.where(lo, m * 2, m)
    k = np.where(lo, e - 1, e).astype(float)
    f = m - 1.0
    s = f / (2.0 + f)
    z = s * s
    w = z * z
    t1 = w * (3.999999999940941908e-01 + w * (2.222219843214978396e-01 + w * 1.531383769920937332e-01))
    t2 = z * (6.666666666666735130e-01 + w * (2.857142874366239149e-01 + w * (1.818357216161805012e-01 + w * 1.479819860511658591e-01)))
    hfsq = 0.5 * f * f
    return k * 6.93147180369123816490e-01 - ((hfsq - (s * (hfsq + (t2 + t1)) + k * 1.90821492927058770002e-10)) - f)


def as241(p):
    q = p - 0.5
    r = 0.180625 - q * q
    cn = (((((((r * 2509.0809287301226727 + 33430.575583588128105) * r + 67265.770927008700853) * r + 45921.953931549871457) * r + 13731.693765509461125) * r + 1971.5909503065514427) * r + 133.14166789178437745) * r + 3.387132872796366608)
    cd = (((((((r * 5226.495278852545925 + 28729.085735721942674) * r + 39307.89580009271061) * r + 21213.794301586595867) * r + 5394.1960214247511077) * r + 687.1870074920579083) * r + 42.313330701600911252) * r + 1.)
    pm = np.where(q < 0, p, 1 - p)
    r = np.sqrt(-flog(pm)) - 1.6
    tn = (((((((r * 7.7454501427834140764e-4 + .0227238449892691845833) * r + .24178072517745061177) * r + 1.27045825245236838258) * r + 3.64784832476320460504) * r + 5.7694972214606914055) * r + 4.6303378461565452959) * r + 1.42343711074968357734)
    td = (((((((r * 1.05075007164441684324e-9 + 5.475938084995344946e-4) * r + .0151986665636164571966) * r + .14810397642748007459) * r + .68976733498510000455) * r + 1.6763848301838038494) * r + 2.05319162663775882187) * r + 1.)
    central = np.abs(q) <= 0.425
    return np.where(central, q * cn, np.where(q < 0, -tn, tn)) / np.where(central, cd, td)


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    p = np.concatenate([rng.random(2_000_000), np.exp(-rng.uniform(0.3, 5, 500_000)**2),
                        1 - np.exp(-rng.uniform(0.3, 5, 500_000)**2)])
    p = p[(p > np.exp(-25.0)) & (p < 1) & (p != 0.5)]
    b = ndtri(p)
    rel = np.abs(as241(p) - b) / np.abs(b)
    x = np.concatenate([rng.random(1_000_000), np.exp(-rng.uniform(0, 30, 1_000_000))])
    x = x[x > 0]
    print("log_pos: max |error| in ulp of ln x:", (np.abs(flog(x) - np.log(x)) / np.spacing(np.abs(np.log(x))))[np.abs(np.log(x)) > 1e-3].max())
    print("points", len(p), "max rel err vs scipy.special.ndtri", rel.max(), "at p =", p[rel.argmax()])
