# cython: language_level=3, boundscheck=False, wraparound=False
"""Cython binding of the MI355X engine's C ABI (include/smcpp_engine.h) with the surface of the reference's
`smcpp/_smcpp.pyx` (122-412): `PyOnePopInferenceManager`, `PyTwoPopInferenceManager`, `PyRateFunction`, `raw_sfs`,
`set_num_threads`, the logger callback / abort flag (32-55) and `_init_cache` (24-30).  This is the file a maintainer
of the reference drops in place of `smcpp/_smcpp.pyx` + `_smcpp.pxd` (INTEGRATION.md); in this repository it is
compiled next to the ctypes binding (`smcpp_amd/_smcpp.py`) and tested against it.

Values that carry derivatives in the reference (`Q()`, `pi`, `transition`, `emission`, `emission_probs`, `R`, ...)
are returned as ad numbers built from the engine's Jacobians in `model.dlist` order (`_adouble_to_ad`,
_smcpp.pyx:103-114)."""
import logging
import os
import sys

import numpy as np
cimport numpy as np
from libc.stdlib cimport malloc, free

try:                                     # inside the reference's package: its own ad numbers
    from smcpp.ad import adnumber, ADF
except ImportError:                      # stand-alone: this repository's
    from smcpp_amd.ad import adnumber, ADF

logger = logging.getLogger(__name__)

# --- begin generated from include/smcpp_engine.h (python -m smcpp_amd._cabi --write) ---
cdef extern from "smcpp_engine.h":
    ctypedef struct smcpp_im:
        pass
    const char *smcpp_last_error() nogil
    int smcpp_create_onepop(int n, int n_contigs, const int *Ls, const int *const *obs, int n_hs, const double *hs, double polarization_error, int device, smcpp_im **out) nogil
    int smcpp_create_twopop(int n1, int n2, int a1, int a2, int n_contigs, const int *Ls, const int *const *obs, int n_hs, const double *hs, double polarization_error, int device, smcpp_im **out) nogil
    void smcpp_destroy(smcpp_im *im) nogil
    int smcpp_set_theta(smcpp_im *im, double theta) nogil
    int smcpp_set_rho(smcpp_im *im, double rho) nogil
    int smcpp_set_alpha(smcpp_im *im, double alpha) nogil
    int smcpp_set_params(smcpp_im *im, int K, const double *a, const double *da, int nder, const double *s) nogil
    int smcpp_set_raw(smcpp_im *im, const double *pi, const double *T, int K, const int *keys, const double *E) nogil
    int smcpp_estep(smcpp_im *im, int forward_backward_only) nogil
    int smcpp_loglik(smcpp_im *im, double *out) nogil
    int smcpp_q(smcpp_im *im, double *val, double *jac) nogil
    int smcpp_num_derivatives(smcpp_im *im) nogil
    int smcpp_set_save_gamma(smcpp_im *im, int on) nogil
    int smcpp_get_save_gamma(smcpp_im *im) nogil
    int smcpp_num_states(smcpp_im *im) nogil
    int smcpp_num_contigs(smcpp_im *im) nogil
    int smcpp_num_keys(smcpp_im *im) nogil
    int smcpp_key_len(smcpp_im *im) nogil
    int smcpp_get_hidden_states(smcpp_im *im, double *hs) nogil
    int smcpp_set_hidden_states(smcpp_im *im, int n_hs, const double *hs) nogil
    int smcpp_get_keys(smcpp_im *im, int *keys) nogil
    int smcpp_get_xisum(smcpp_im *im, int contig, double *out) nogil
    int smcpp_get_gamma(smcpp_im *im, int contig, double *out) nogil
    int smcpp_gamma_cols(smcpp_im *im, int contig) nogil
    int smcpp_get_gamma_sums(smcpp_im *im, int contig, double *vals, unsigned char *present) nogil
    int smcpp_get_pi(smcpp_im *im, double *out) nogil
    int smcpp_get_transition(smcpp_im *im, double *out) nogil
    int smcpp_get_emission_probs(smcpp_im *im, double *out) nogil
    int smcpp_get_pi_jac(smcpp_im *im, double *out) nogil
    int smcpp_get_transition_jac(smcpp_im *im, double *out) nogil
    int smcpp_get_emission_probs_jac(smcpp_im *im, double *out) nogil
    int smcpp_num_emission_cols(smcpp_im *im) nogil
    int smcpp_get_emission(smcpp_im *im, double *out, double *jac) nogil
    int smcpp_get_gamma_argmax(smcpp_im *im, int contig, int *out) nogil
    int smcpp_set_global_keys(smcpp_im *im, int Kg, const int *gkeys) nogil
    int smcpp_pack_stats(smcpp_im *im, double *buf, long *n_out, int dev) nogil
    int smcpp_unpack_stats(smcpp_im *im, const double *buf, long n, int dev) nogil
    int smcpp_rccl_unique_id(const char *libpath, char *out128) nogil
    int smcpp_rccl_init(smcpp_im *im, const char *libpath, const char *id128, int rank, int world) nogil
    int smcpp_rccl_exchange(smcpp_im *im, double *loglik_sum) nogil
    int smcpp_rccl_unpack(smcpp_im *im) nogil
    int smcpp_rccl_fetch(smcpp_im *im, double *out, long n) nogil
    int smcpp_rccl_destroy(smcpp_im *im) nogil
    int smcpp_set_debug(smcpp_im *im, int on) nogil
    int smcpp_get_debug(smcpp_im *im) nogil
    int smcpp_set_chunking(smcpp_im *im, int rows_per_chunk, double eps_alpha, double eps_beta) nogil
    int smcpp_set_prep_mode(smcpp_im *im, int host) nogil
    int smcpp_set_warm_start(smcpp_im *im, int on) nogil
    int smcpp_last_timing(smcpp_im *im, double *out) nogil
    int smcpp_host_chunk_counts(int n_contigs, const long long *cost, const int *rows, long long nslots, long long floor_cost, int *out) nogil
    int smcpp_chain_mode(smcpp_im *im) nogil
    void smcpp_reload_options() nogil
    int smcpp_describe(smcpp_im *im, char *buf, int cap) nogil
    int smcpp_debug_ss_apply(int M, const double *T, int nvec, const double *x, const double *e, double *out_f, double *out_b) nogil
    int smcpp_debug_ss_apply_float_scans(int M, const double *T, int nvec, const double *x, const double *e, double *out_f, double *out_b) nogil
    int smcpp_last_host_timing(smcpp_im *im, double *out) nogil
    void *smcpp_stream(smcpp_im *im) nogil
    int smcpp_device(smcpp_im *im) nogil
    void smcpp_init_logger_cb(void (*cb)(const char *, const char *, const char *)) nogil
    int smcpp_init_cache(const char *path) nogil
    void smcpp_set_num_threads(int k) nogil
    int smcpp_dev_shape(int mode, long long L, int ncol, const int *rows, long long p0, long long p1, const long long *na, long long *rows_out, double *kernel_ms) nogil
    int smcpp_dev_shape_fetch(int *out) nogil
    int smcpp_host_set_csfs_direct(int on) nogil
    int smcpp_host_eigensystem(int n, const double *A, double *P, double *Pinv, double *d, double *scale, double *max_imag) nogil
    int smcpp_host_eigensystem_team(int n, const double *A, int threads, double *P, double *Pinv, double *d, double *scale, double *max_imag) nogil
    int smcpp_host_prep_onepop(int n, int n_hs, const double *hs, double polarization_error, int Kp, const double *a, const double *s, double theta, double rho, double alpha, int K, const int *keys, double *pi, double *T, double *E) nogil
    int smcpp_host_prep_onepop_jac(int n, int n_hs, const double *hs, double polarization_error, int Kp, const double *a, const double *da, int nder, const double *s, double theta, double rho, double alpha, int K, const int *keys, double *pi, double *T, double *E, double *dpi, double *dT, double *dE) nogil
    int smcpp_dev_prep_onepop(int n, int n_hs, const double *hs, double polarization_error, int Kp, const double *a, const double *da, int nder, const double *s, double theta, double rho, double alpha, int K, const int *keys, int mode, double *pi, double *T, double *E, double *dpi, double *dT, double *dE, double *sfs, double *dsfs) nogil
    int smcpp_dev_q_emulate(int n, int n_hs, const double *hs, double polarization_error, int Kp, const double *a, const double *da, int nder, const double *s, double theta, double rho, double alpha, int K, const int *keys, const double *g0, const double *xi, const double *gs, double *val, double *jac) nogil
    int smcpp_host_rate_function(int Kp, const double *a, const double *s, int n_hs, const double *hs, int nt, const double *t, double *R_out, double *avg_ct_out) nogil
    int smcpp_host_rate_function_jac(int Kp, const double *a, const double *da, int nder, const double *s, int n_hs, const double *hs, int nt, const double *t, double *R_out, double *dR_out, double *avg_ct_out, double *davg_ct_out) nogil
    int smcpp_host_random_coal_times(int Kp, const double *a, const double *s, double t1, double t2, int K, const unsigned long long *seeds, double *t_out, double *R_out) nogil
    int smcpp_host_raw_sfs(int n, int Kp, const double *a, const double *da, int nder, const double *s, double t1, double t2, int below_only, double *sfs, double *dsfs) nogil
    int smcpp_set_params_twopop(smcpp_im *im, int Kd, const double *ad, const double *sd, const double *dad, int K1, const double *a1, const double *s1, const double *da1, int K2, const double *a2, const double *s2, const double *da2, double split, int nder) nogil
    int smcpp_host_joint_csfs(int n1, int n2, int a1, int a2, int n_hs, const double *hs, int K1, const double *pa1, const double *ps1, const double *da1, int K2, const double *pa2, const double *ps2, const double *da2, int nder, double split, int Kmc, double *out, double *dout) nogil
    int smcpp_host_prep_twopop(int n1, int n2, int a1, int a2, int n_hs, const double *hs, double polarization_error, int Kd, const double *ad, const double *sd, int K1, const double *pa1, const double *ps1, int K2, const double *pa2, const double *ps2, double split, double theta, double rho, double alpha, int K, const int *keys, double *pi, double *T, double *E) nogil
# --- end generated ---

aca = np.ascontiguousarray


def _init_cache():
    """`_smcpp.pyx:24-30`: the on-disk store of the n-only tables lives in the user cache directory."""
    base = os.environ.get("XDG_CACHE_HOME", os.path.join(os.path.expanduser("~"), ".cache"))
    d = os.path.join(base, "smcpp_amd")
    try:
        os.makedirs(d, exist_ok=True)
    except OSError:
        return
    smcpp_init_cache(os.path.join(d, "matrices.dat").encode("UTF-8"))


# Engine messages -> Python `logging`.  Contract kept from the reference (_smcpp.pyx:32-55, _smcpp.pxd:26): the native side may
# fire the callback from any host thread; records appear under the logger namespace the reference's users configure; a Ctrl-C that
# lands inside the callback must not unwind through C++ frames, so it is parked in the module flag `abort` and re-raised by
# `_check_abort()`, which every heavy call runs once it has returned.
abort = False
_LOG_NAMESPACE = "smcpp._smcpp"
_LEVELS_BELOW_DEBUG = {"DEBUG1": logging.DEBUG - 1}


def _level_number(text):
    text = text.upper()
    if text in _LEVELS_BELOW_DEBUG:
        return _LEVELS_BELOW_DEBUG[text]
    number = logging.getLevelName(text)          # the registered number of a level name (a string back for an unknown one)
    return number if isinstance(number, int) else logging.INFO


cdef void _forward_log(const char *name, const char *level, const char *message) noexcept with gil:
    global abort
    target = logging.getLogger("%s:%s" % (_LOG_NAMESPACE, name.decode("utf-8", "replace")))
    try:
        target.log(_level_number(level.decode("utf-8", "replace")), "%s", message.decode("utf-8", "replace"))
    except KeyboardInterrupt:
        abort = True
        target.critical("Aborting")


def _check_abort():
    global abort
    pending, abort = abort, False
    if pending:
        raise KeyboardInterrupt()


smcpp_init_logger_cb(_forward_log)


def set_num_threads(k):
    smcpp_set_num_threads(int(k))


cdef _check(int rc):
    if rc != 0:
        raise RuntimeError(smcpp_last_error().decode("UTF-8"))


def _params(a, s, dlist):
    """`make_params` (_smcpp.pyx:66-83): piece sizes (floats or ad numbers) -> values [K], seeds [K x nder], lengths."""
    a = list(a)
    assert len(a) > 0
    vals = aca(np.array([aa.x if isinstance(aa, ADF) else float(aa) for aa in a], dtype=np.float64))
    assert np.all(vals > 0)
    nder = len(dlist)
    da = np.zeros((len(a), max(nder, 1)))
    for k, aa in enumerate(a):
        if isinstance(aa, ADF):
            for j, dv in enumerate(dlist):
                da[k, j] = aa.d(dv)
    return vals, aca(da), nder, aca(np.array([float(x) for x in s], dtype=np.float64))


def _to_ad(double x, jac, dlist):
    """`_adouble_to_ad` (_smcpp.pyx:103-114)."""
    if len(dlist) == 0:
        return x
    r = adnumber(x)
    dd = r.d()
    for i, dv in enumerate(dlist):
        dd[dv] = float(jac[i])
    return r


def _ad_array(vals, jac, dlist):
    vals = np.asarray(vals)
    out = np.zeros(vals.shape, dtype=object)
    if len(dlist) == 0:
        out[...] = vals
        return out
    flat_v = vals.reshape(-1)
    flat_j = np.asarray(jac).reshape(len(flat_v), -1)
    flat_o = out.reshape(-1)
    for i in range(len(flat_v)):
        flat_o[i] = _to_ad(flat_v[i], flat_j[i], dlist)
    return out


cdef class _PyInferenceManager:
    cdef smcpp_im *_im
    cdef int _num_hmms
    cdef object _model, _observations, _theta, _rho, _alpha, _im_id, _Ls, _hs, _keep
    cdef public long long seed
    cdef object __weakref__

    def _my_init(self, observations, hidden_states, im_id=None):
        _init_cache()
        self._im_id = im_id
        self.seed = 1
        if len(observations) == 0:
            raise RuntimeError("Observations list is empty")
        hidden_states = np.asarray(hidden_states, dtype=np.float64)
        if not np.all(np.sort(hidden_states) == hidden_states):
            raise RuntimeError("Hidden states must be in ascending order")
        self._observations = observations
        self._keep = [aca(ob, dtype=np.int32) for ob in observations]
        self._Ls = aca(np.array([ob.shape[0] for ob in self._keep], dtype=np.int32))
        self._hs = aca(hidden_states)
        self._num_hmms = len(observations)
        _check_abort()

    def __dealloc__(self):
        if self._im != NULL:
            smcpp_destroy(self._im)
            self._im = NULL

    cdef _dlist(self):
        return list(self._model.dlist) if self._model is not None else []

    property observations:
        def __get__(self):
            return self._observations

    property theta:
        def __get__(self):
            return self._theta

        def __set__(self, theta):
            self._theta = theta
            _check(smcpp_set_theta(self._im, theta))

    property rho:
        def __get__(self):
            return self._rho

        def __set__(self, rho):
            self._rho = rho
            _check(smcpp_set_rho(self._im, rho))

    property alpha:
        def __get__(self):
            return self._alpha

        def __set__(self, alpha):
            self._alpha = alpha
            _check(smcpp_set_alpha(self._im, alpha))

    def E_step(self, forward_backward_only=False):
        if None in (self.theta, self.rho, self.alpha):
            raise RuntimeError("theta / rho / alpha must be set")
        cdef int fb = 1 if forward_backward_only else 0
        cdef int rc
        with nogil:
            rc = smcpp_estep(self._im, fb)
        _check(rc)
        _check_abort()

    property model:
        def __get__(self):
            return self._model

        def __set__(self, m):
            self._model = m
            m.register(self)
            self.update("model update")

    property save_gamma:
        def __get__(self):
            return bool(smcpp_get_save_gamma(self._im))

        def __set__(self, bint sg):
            _check(smcpp_set_save_gamma(self._im, sg))

    property debug:
        # InferenceManager::debug (_smcpp.pxd:53)
        def __get__(self):
            return bool(smcpp_get_debug(self._im))

        def __set__(self, bint on):
            _check(smcpp_set_debug(self._im, on))

    property hidden_states:
        def __get__(self):
            cdef np.ndarray[double, ndim=1] hs = np.zeros(len(self._hs))
            _check(smcpp_get_hidden_states(self._im, &hs[0]))
            return list(hs)

        def __set__(self, hs):
            cdef np.ndarray[double, ndim=1] h = aca(hs, dtype=np.float64)
            if len(h) != len(self._hs):
                raise RuntimeError("hidden states must be same size")
            _check(smcpp_set_hidden_states(self._im, len(h), &h[0]))

    def _keys(self):
        cdef int K = smcpp_num_keys(self._im), kl = smcpp_key_len(self._im)
        cdef np.ndarray[int, ndim=2] k = np.zeros((K, kl), dtype=np.int32)
        _check(smcpp_get_keys(self._im, &k[0, 0]))
        return k

    property emission_probs:
        def __get__(self):
            keys = self._keys()
            cdef int M = len(self._hs) - 1, K = len(keys), nder = smcpp_num_derivatives(self._im)
            cdef np.ndarray[double, ndim=2] out = np.zeros((K, M))
            cdef np.ndarray[double, ndim=3] jac = np.zeros((K, M, max(nder, 1)))
            _check(smcpp_get_emission_probs(self._im, &out[0, 0]))
            dlist = self._dlist()
            if nder:
                _check(smcpp_get_emission_probs_jac(self._im, &jac[0, 0, 0]))
            return {tuple(int(x) for x in keys[k]): _ad_array(out[k], jac[k], dlist if nder else []) for k in range(K)}

    property gamma_sums:
        def __get__(self):
            keys = self._keys()
            cdef int M = len(self._hs) - 1, K = len(keys), c
            cdef np.ndarray[double, ndim=2] vals
            cdef np.ndarray[unsigned char, ndim=1] present
            ret = []
            for c in range(self._num_hmms):
                vals = np.zeros((K, M))
                present = np.zeros(K, dtype=np.uint8)
                _check(smcpp_get_gamma_sums(self._im, c, &vals[0, 0], &present[0]))
                ret.append({tuple(int(x) for x in keys[k]): vals[k].copy() for k in range(K) if present[k]})
            return ret

    property gammas:
        def __get__(self):
            cdef int M = len(self._hs) - 1, c
            cdef np.ndarray[double, ndim=2] g
            ret = []
            for c in range(self._num_hmms):
                g = np.zeros((M, smcpp_gamma_cols(self._im, c)))
                _check(smcpp_get_gamma(self._im, c, &g[0, 0]))
                ret.append(g)
            return ret

    property xisums:
        def __get__(self):
            cdef int M = len(self._hs) - 1, c
            cdef np.ndarray[double, ndim=2] x
            ret = []
            for c in range(self._num_hmms):
                x = np.zeros((M, M))
                _check(smcpp_get_xisum(self._im, c, &x[0, 0]))
                ret.append(x)
            return ret

    property pi:
        def __get__(self):
            cdef int M = len(self._hs) - 1, nder = smcpp_num_derivatives(self._im)
            cdef np.ndarray[double, ndim=1] v = np.zeros(M)
            cdef np.ndarray[double, ndim=2] j = np.zeros((M, max(nder, 1)))
            _check(smcpp_get_pi(self._im, &v[0]))
            if nder:
                _check(smcpp_get_pi_jac(self._im, &j[0, 0]))
            return _ad_array(v.reshape(M, 1), j, self._dlist() if nder else [])       # Matrix<adouble> M x 1

    property transition:
        def __get__(self):
            cdef int M = len(self._hs) - 1, nder = smcpp_num_derivatives(self._im)
            cdef np.ndarray[double, ndim=2] v = np.zeros((M, M))
            cdef np.ndarray[double, ndim=3] j = np.zeros((M, M, max(nder, 1)))
            _check(smcpp_get_transition(self._im, &v[0, 0]))
            if nder:
                _check(smcpp_get_transition_jac(self._im, &j[0, 0, 0]))
            return _ad_array(v, j, self._dlist() if nder else [])

    property emission:
        def __get__(self):
            cdef int M = len(self._hs) - 1, nder = smcpp_num_derivatives(self._im), cols = smcpp_num_emission_cols(self._im)
            cdef np.ndarray[double, ndim=2] v = np.zeros((M, cols))
            cdef np.ndarray[double, ndim=3] j = np.zeros((M, cols, max(nder, 1)))
            _check(smcpp_get_emission(self._im, &v[0, 0], &j[0, 0, 0]))
            return _ad_array(v, j, self._dlist() if nder else [])

    def Q(self, separate=False):
        cdef int nder = smcpp_num_derivatives(self._im), rc
        cdef np.ndarray[double, ndim=1] q = np.zeros(4)
        cdef np.ndarray[double, ndim=2] jac = np.zeros((4, max(nder, 1)))
        cdef double *qp = &q[0]
        cdef double *jp = &jac[0, 0]
        with nogil:
            rc = smcpp_q(self._im, qp, jp)
        if rc != 0:
            msg = smcpp_last_error().decode("UTF-8")
            if msg in ("SFS is not a probability distribution", "csfs is not a probability distribution"):
                logger.warning("Model does not induce a valid probability distribution")
                return adnumber(-np.inf)
            raise RuntimeError(msg)
        _check_abort()
        dlist = self._dlist() if nder else []
        qq = [_to_ad(q[i], jac[i], dlist) for i in range(4)]
        for i in range(4):
            logger.debug("im(%r).q%d: %s", self._im_id, i + 1, qq[i])
        if separate:
            return qq
        if not dlist:
            return adnumber(float(q.sum()))
        return _to_ad(float(q.sum()), jac.sum(axis=0), dlist)

    def loglik(self):
        cdef np.ndarray[double, ndim=1] ll = np.zeros(self._num_hmms)
        cdef double *lp = &ll[0]
        cdef int rc
        with nogil:
            rc = smcpp_loglik(self._im, lp)
        _check(rc)
        _check_abort()
        return float(ll.sum())


cdef class PyOnePopInferenceManager(_PyInferenceManager):

    def __cinit__(self, int n, observations, hidden_states, im_id, double polarization_error, int device=-1):
        self._my_init(observations, hidden_states, im_id)
        cdef np.ndarray[int, ndim=1] Ls = self._Ls
        cdef np.ndarray[double, ndim=1] hs = self._hs
        cdef const int **ptrs = <const int **>malloc(sizeof(void *) * self._num_hmms)
        cdef np.ndarray[int, ndim=2] ob
        for i in range(self._num_hmms):
            ob = self._keep[i]
            ptrs[i] = &ob[0, 0]
        cdef int rc
        cdef int nh = self._num_hmms, nhs = len(hs)
        with nogil:
            rc = smcpp_create_onepop(n, nh, &Ls[0], ptrs, nhs, &hs[0], polarization_error, device, &self._im)
        free(<void *>ptrs)
        _check(rc)
        self.alpha = 1
        self.theta = 1e-4
        self.rho = 1e-4

    @property
    def pid(self):
        assert len(self._im_id) == 1
        return self._im_id[0]

    def update(self, message, *args, **kwargs):
        m = self._model.for_pop(self.pid)
        a, da, nder, s = _params(m.stepwise_values(), m.s, m.dlist)
        cdef np.ndarray[double, ndim=1] av = a, sv = s
        cdef np.ndarray[double, ndim=2] dv = da
        cdef int K = len(a), nd = nder, rc
        with nogil:
            rc = smcpp_set_params(self._im, K, &av[0], &dv[0, 0] if nd > 0 else NULL, nd, &sv[0])
        _check(rc)


cdef class PyTwoPopInferenceManager(_PyInferenceManager):
    cdef int _a1

    def __cinit__(self, int n1, int n2, int a1, int a2, observations, hidden_states, im_id, double polarization_error,
                  int device=-1):
        assert a1 + a2 == 2
        assert a1 in [1, 2]
        assert a2 in [0, 1]
        self._a1 = a1
        self._my_init(observations, hidden_states, im_id)
        cdef np.ndarray[int, ndim=1] Ls = self._Ls
        cdef np.ndarray[double, ndim=1] hs = self._hs
        cdef const int **ptrs = <const int **>malloc(sizeof(void *) * self._num_hmms)
        cdef np.ndarray[int, ndim=2] ob
        for i in range(self._num_hmms):
            ob = self._keep[i]
            ptrs[i] = &ob[0, 0]
        cdef int rc
        cdef int nh = self._num_hmms, nhs = len(hs)
        with nogil:
            rc = smcpp_create_twopop(n1, n2, a1, a2, nh, &Ls[0], ptrs, nhs, &hs[0], polarization_error, device, &self._im)
        free(<void *>ptrs)
        _check(rc)
        self.alpha = 1
        self.theta = 1e-4
        self.rho = 1e-4

    def update(self, message, *args, **kwargs):
        m = self._model
        pids = self._im_id
        dist = None if self._a1 == 1 else pids[0]
        dm = m.for_pop(dist)
        ms = [m.for_pop(p) for p in pids]
        ad_, dad, nder, sd = _params(dm.stepwise_values(), dm.s, m.dlist)
        a1, da1, _, s1 = _params(ms[0].stepwise_values(), ms[0].s, m.dlist)
        a2, da2, _, s2 = _params(ms[1].stepwise_values(), ms[1].s, m.dlist)
        cdef np.ndarray[double, ndim=1] adv = ad_, sdv = sd, a1v = a1, s1v = s1, a2v = a2, s2v = s2
        cdef np.ndarray[double, ndim=2] dadv = dad, da1v = da1, da2v = da2
        cdef double split = m.split
        cdef int nd = nder, rc
        with nogil:
            rc = smcpp_set_params_twopop(self._im, <int>adv.shape[0], &adv[0], &sdv[0], &dadv[0, 0] if nd > 0 else NULL,
                                         <int>a1v.shape[0], &a1v[0], &s1v[0], &da1v[0, 0] if nd > 0 else NULL,
                                         <int>a2v.shape[0], &a2v[0], &s2v[0], &da2v[0, 0] if nd > 0 else NULL, split, nd)
        _check(rc)


cdef class PyRateFunction:
    cdef object _model, _a, _da, _s, _hs
    cdef int _nder

    def __cinit__(self, model, hs):
        self._model = model
        self._a, self._da, self._nder, self._s = _params(model.stepwise_values(), model.s, model.dlist)
        self._hs = aca(np.array(list(hs), dtype=np.float64))

    def _eval(self, t):
        cdef np.ndarray[double, ndim=1] a = self._a, s = self._s, hs = self._hs
        cdef np.ndarray[double, ndim=2] da = self._da
        cdef np.ndarray[double, ndim=1] tt = aca(np.atleast_1d(t), dtype=np.float64)
        cdef int nt = len(tt), nhs = len(hs), nd = self._nder
        cdef np.ndarray[double, ndim=1] R = np.zeros(max(nt, 1)), ct = np.zeros(max(nhs - 1, 1))
        cdef np.ndarray[double, ndim=2] dR = np.zeros((max(nt, 1), max(nd, 1))), dct = np.zeros((max(nhs - 1, 1), max(nd, 1)))
        _check(smcpp_host_rate_function_jac(len(a), &a[0], &da[0, 0], nd, &s[0], nhs if nhs >= 2 else 0,
                                            &hs[0] if nhs >= 2 else NULL, nt, &tt[0] if nt else NULL, &R[0], &dR[0, 0],
                                            &ct[0] if nhs >= 2 else NULL, &dct[0, 0] if nhs >= 2 else NULL))
        return R, dR, ct, dct

    def R(self, t):
        assert np.isfinite(t)
        R, dR, _, _ = self._eval([float(t)])
        return _to_ad(R[0], dR[0], list(self._model.dlist))

    def average_coal_times(self):
        if len(self._hs) < 2:
            return []
        _, _, ct, dct = self._eval([])
        dl = list(self._model.dlist)
        return [_to_ad(ct[i], dct[i], dl) for i in range(len(self._hs) - 1)]

    def random_coal_times(self, t1, t2, K):
        cdef np.ndarray[double, ndim=1] a = self._a, s = self._s
        cdef np.ndarray[np.uint64_t, ndim=1] seeds = np.random.randint(0, sys.maxsize, size=K, dtype=np.int64).astype(np.uint64)
        cdef np.ndarray[double, ndim=1] t = np.zeros(K), R = np.zeros(K)
        _check(smcpp_host_random_coal_times(len(a), &a[0], &s[0], t1, t2, K, <const unsigned long long *>&seeds[0], &t[0], &R[0]))
        return [[float(t[i]), float(R[i])] for i in range(K)]


def raw_sfs(model, int n, double t1, double t2, below_only=False):
    a, da, nder, s = _params(model.stepwise_values(), model.s, model.dlist)
    cdef np.ndarray[double, ndim=1] av = a, sv = s
    cdef np.ndarray[double, ndim=2] dv = da
    cdef np.ndarray[double, ndim=2] out = np.zeros((3, n + 1))
    cdef np.ndarray[double, ndim=3] dout = np.zeros((3, n + 1, max(nder, 1)))
    _check(smcpp_host_raw_sfs(n, len(a), &av[0], &dv[0, 0] if nder else NULL, nder, &sv[0], t1, t2, 1 if below_only else 0,
                              &out[0, 0], &dout[0, 0, 0] if nder else NULL))
    _check_abort()
    return _ad_array(out, dout, list(model.dlist) if nder else [])


# Used for testing purposes only
def joint_csfs(int n1, int n2, int a1, int a2, model, hidden_states, int K=10):
    """`joint_csfs` of the reference (smcpp/_smcpp.pyx:416-437): per hidden state the joint conditioned SFS
    [(a1 + 1) x (n1 + 1) x (a2 + 1) x (n2 + 1)] of a two-population model (JointCSFS, src/jcsfs.cpp:219-420), entries as ad
    numbers in `model.dlist` order when the model carries derivatives."""
    assert (a1 == 2 and a2 == 0) or (a1 == a2 == 1)
    dl = list(model.dlist)
    a1v, d1, nder, s1v = _params(model.model1.stepwise_values(), model.model1.s, dl)
    a2v, d2, _nd2, s2v = _params(model.model2.stepwise_values(), model.model2.s, dl)
    cdef np.ndarray[double, ndim=1] hs = aca(np.array(list(hidden_states), dtype=np.float64))
    cdef np.ndarray[double, ndim=1] pa1 = a1v, ps1 = s1v, pa2 = a2v, ps2 = s2v
    cdef np.ndarray[double, ndim=2] da1 = d1, da2 = d2
    cdef int M = len(hs) - 1, nd = nder
    cdef long sz = (a1 + 1) * (n1 + 1) * (a2 + 1) * (n2 + 1)
    cdef np.ndarray[double, ndim=2] out = np.zeros((M, sz))
    cdef np.ndarray[double, ndim=3] dout = np.zeros((M, sz, max(nd, 1)))
    cdef double split = float(model.split)
    cdef int rc
    with nogil:
        rc = smcpp_host_joint_csfs(n1, n2, a1, a2, M + 1, &hs[0], pa1.shape[0], &pa1[0], &ps1[0], &da1[0, 0] if nd else NULL,
                                   pa2.shape[0], &pa2[0], &ps2[0], &da2[0, 0] if nd else NULL, nd, split, K, &out[0, 0],
                                   &dout[0, 0, 0] if nd else NULL)
    _check(rc)
    _check_abort()
    ret = []
    for i in range(M):
        mat = _ad_array(out[i], dout[i], dl if nd else [])
        mat = np.asarray(mat).reshape(a1 + 1, n1 + 1, a2 + 1, n2 + 1)
        ret.append(mat)
    return ret
