"""TEST INFRASTRUCTURE ONLY — JointCSFS for one distinguished lineage per population (a1 = a2 = 1).

The reference implements this configuration only in C++ (``JointCSFS<T>::pre_compute_apart``, src/jcsfs.cpp:258-367;
the Python original ``smcpp/jcsfs.py`` asserts a1 == 2), and jcsfs.cpp needs GSL headers this image does not have, so
that translation unit cannot be built here.  Every numerical building block it calls IS compiled from the reference
sources in ``oracle/_ref`` and is used here through ``oracle/ref.py``:

    shiftParams / truncateParams                src/common.cpp:63-98            ref.shift_or_truncate
    OnePopConditionedSFS(n).compute(eta)        src/conditioned_sfs.cpp:86-97   ref.raw_csfs
    PiecewiseConstantRateFunction::R            ...rate_function.cpp:406-411    ref.raw_csfs(t=...)
    modified_moran_rate_matrix(N, a, 1)         src/moran_eigensystem.cpp:31-52 ref.modified_moran

What is RESTATED here (and therefore pinned only by agreement of two independent restatements, this one in numpy
and the product's in C++): the assembly loops of jcsfs.cpp:258-367 themselves, ``undistinguishedSFS`` (56-67), the
threshold / zeroing epilogue of ``compute`` (228-243), the matrix exponential of the rate matrices (scipy.linalg.expm
instead of the eigen-decomposition of include/jcsfs.h ``jcsfs_eigensystem::expM``) and the hypergeometric weights
(scipy.stats.hypergeom instead of gsl_ran_hypergeometric_pdf, src/jcsfs.cpp:8-16)."""
import numpy as np
import scipy.linalg
import scipy.stats

from . import ref


def undistinguished_sfs(csfs):
    n = csfs.shape[1] - 1
    ret = np.zeros(n + 1)
    for a in range(3):
        for b in range(n + 1):
            if 1 <= a + b < n + 2:
                ret[a + b - 1] += csfs[a, b]
    return ret


def joint_csfs_apart(n1, n2, hs, model1, model2, split):
    """[M, 2, n1+1, 2, n2+1]; ``model*`` = (a, s) arrays."""
    (a1, s1), (a2, s2) = model1, model2
    hs = np.asarray(hs, dtype=float)
    M = len(hs) - 1
    J = np.zeros((M, 2, n1 + 1, 2, n2 + 1))
    times = [0.0] + [float(t - split) for t in hs[1:M] if t > split] + [np.inf]
    sa, ss = ref.shift_or_truncate(a1, s1, split)
    csfs_at_split, _ = ref.raw_csfs(sa, ss, times, n1 + n2)
    _, R1 = ref.raw_csfs(a1, s1, [0.0, np.inf], -1, t=[split])
    _, R2 = ref.raw_csfs(a2, s2, [0.0, np.inf], -1, t=[split])
    T10 = scipy.linalg.expm(ref.modified_moran(n1, 0, 1) * R1[0])
    T11 = scipy.linalg.expm(ref.modified_moran(n1, 1, 1) * R1[0])
    T20 = scipy.linalg.expm(ref.modified_moran(n2, 0, 1) * R2[0])
    T21 = scipy.linalg.expm(ref.modified_moran(n2, 1, 1) * R2[0])
    i = 0
    for m in range(M):
        if hs[m + 1] <= split:
            continue
        cs = csfs_at_split[i]
        i += 1
        for nseg in range(n1 + n2 + 1):
            for np1 in range(max(nseg - n2, 0), min(nseg, n1) + 1):
                np2 = nseg - np1
                h = scipy.stats.hypergeom.pmf(np1, n1 + n2, nseg, n1)
                J[m, 1, :, 1, :] += h * cs[2, nseg] * np.outer(T11[np1], T21[np2])
                J[m, 1, :, 0, :] += 0.5 * h * cs[1, nseg] * np.outer(T11[np1], T20[np2])
                J[m, 0, :, 1, :] += 0.5 * h * cs[1, nseg] * np.outer(T10[np1], T21[np2])
                J[m, 0, :, 0, :] += h * cs[0, nseg] * np.outer(T10[np1], T20[np2])
    if split != 0.0:
        for first, (a, s, ni) in ((True, (a1, s1, n1)), (False, (a2, s2, n2))):
            ta, ts = ref.shift_or_truncate(a, s, split, truncate=True)
            rsfs = None
            if ni > 0:
                c, _ = ref.raw_csfs(ta, ts, [0.0, np.inf], ni - 1)
                rsfs = undistinguished_sfs(c[0])
            for k in range(1, ni + 1):
                fac = k / (ni + 1.0)
                x1, x2 = (1.0 - fac) * rsfs[k - 1], fac * rsfs[k - 1]
                if first:
                    J[:, 0, k, 0, 0] += x1
                    J[:, 1, k - 1, 0, 0] += x2
                else:
                    J[:, 0, 0, 0, k] += x1
                    J[:, 0, 0, 1, k - 1] += x2
            remain = 0.0
            if ni > 0:
                remain = float(np.arange(1, ni + 1) @ rsfs)
            remain = remain / (ni + 1.0) - split
            if first:
                J[:, 1, ni, 0, 0] -= remain
            else:
                J[:, 0, 0, 1, ni] -= remain
    # JointCSFS::compute epilogue (src/jcsfs.cpp:228-243)
    J = np.where(J > 1e-20, J, 1e-20)
    J[:, 0, 0, 0, 0] = 0.0
    J[:, 1, n1, 1, n2] = 0.0
    return J
