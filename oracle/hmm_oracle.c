/* TEST INFRASTRUCTURE ONLY — plain-C restatement of the reference's per-contig HMM E-step.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product
 * (smcpp_amd/) never does.  It is pinned against the compiled reference (oracle/_ref, see oracle/ref.py) by
 * tests/test_oracle_vs_ref.py and against the committed golden vectors under tests/golden/.
 *
 * What is restated (reference file:line):
 *   oracle_span_q      TransitionBundle::update, second loop           src/transition_bundle.cpp:29-59
 *   oracle_estep       HMM::Estep                                      src/hmm.cpp:45-153
 *   oracle_q           HMM::Q (values only, no derivatives)            src/hmm.cpp:155-193
 *                      doubly_compensated_summation                    include/common.h:27-46
 * The eigendecomposition of diag(b_k) * Td^T (transition_bundle.cpp:15-25, transition_bundle.h:9-30) is done by
 * the caller (oracle/oracle.py, LAPACK via numpy) and handed in as P_r, Pinv_r, d_r, scale per key.
 *
 * Quirks mirrored on purpose (SURVEY.md §8(a) quirk ledger): alpha_hat is float; the span-1 forward mat-vec is a
 * float AXPY chain over k without FMA; the float sum uses Eigen's 2x4-lane packet order when M % 4 == 0; the
 * 1e-10f clamp is not renormalised; log_c uses the pre-clamp normaliser; eigen rows use unscaled d_r in the
 * gamma diagonal but scaled d^span in beta; gamma[:,0] is not normalised; xisum gets o Td and the 1e-20 floor
 * once at the end.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define IDX(i, j, n) ((size_t)(i) * (size_t)(n) + (size_t)(j))

/* span_Qs entry for one (span, eigenvalues) pair.  S is M x M row-major. */
void oracle_span_q(int M, int span, const double *d_scaled, double *S)
{
    for (int a = 0; a < M; ++a)
    {
        double d1 = d_scaled[a];
        S[IDX(a, a, M)] = pow(d1, span - 1) * (double)span;
        for (int b = a + 1; b < M; ++b)
        {
            d1 = d_scaled[a];
            double d2 = d_scaled[b];
            if (fabs(d1) < fabs(d2)) { double t = d1; d1 = d2; d2 = t; }
            double q = exp((double)span * log(d1) + log1p(-pow(d2 / d1, span)));
            q /= d1 - d2;
            S[IDX(a, b, M)] = q;
            S[IDX(b, a, M)] = q;
        }
    }
}

/* Eigen 3.3.3 redux order for a 16-byte aligned float column with M % 4 == 0. */
static float eigen_float_sum(const float *x, int M)
{
    if (M % 4 != 0 || M < 4)
    {
        float s = 0.f;
        for (int i = 0; i < M; ++i) s += x[i];
        return s;
    }
    int npk = M / 4;
    float p0[4], p1[4];
    memcpy(p0, x, sizeof p0);
    if (npk >= 2)
    {
        memcpy(p1, x + 4, sizeof p1);
        int k = 2;
        for (; k + 1 < npk; k += 2)
            for (int j = 0; j < 4; ++j)
            {
                p0[j] += x[4 * k + j];
                p1[j] += x[4 * (k + 1) + j];
            }
        for (int j = 0; j < 4; ++j) p0[j] += p1[j];
        if (k < npk)
            for (int j = 0; j < 4; ++j) p0[j] += x[4 * k + j];
    }
    return (p0[0] + p0[2]) + (p0[1] + p0[3]);
}

static double vsum(const double *x, int n)
{
    double s = 0.;
    for (int i = 0; i < n; ++i) s += x[i];
    return s;
}

/* One contig.
 *   E [K x M] emission vector per key id; pi [M]; T [M x M] row-major (Td);
 *   span[L], kid[L]: row ell-1 has span[ell-1] and key id kid[ell-1];
 *   eig_of_key[K]: index into the eigensystem arrays or -1; P, Pinv [nEig x M x M] row-major; d [nEig x M]
 *   (unscaled real parts); scale[nEig].
 * Outputs: loglik[1]; alpha_hat [(L+1) x M] float; log_c [L+1]; xisum [M x M] row-major;
 *   gamma: [(L+1) x M] if save_gamma (row ell = reference column ell) else [M] (= reference gamma.col(0));
 *   gamma_sums [K x M] and gs_present[K] (1 if the reference's map would hold the key);
 *   beta_store (optional, [(L+1) x M]): row ell = the beta vector the backward loop holds when it processes
 *   row ell (row L = ones), row 0 = the final normalised beta.
 * Returns 0, or 1 if a span>1 row has no eigensystem / a span-1 branch is hit with span != 1 (hmm.cpp:132-133).
 */
int oracle_estep(int M, int K, const double *E, const double *pi, const double *T,
                 int L, const int *span, const int *kid,
                 const int *eig_of_key, const double *P, const double *Pinv, const double *d, const double *scale,
                 int save_gamma,
                 double *loglik, float *alpha_hat, double *log_c, double *xisum, double *gamma,
                 double *gamma_sums, unsigned char *gs_present, double *beta_store)
{
    const size_t MM = (size_t)M * M;
    double *a = malloc(sizeof(double) * M), *u = malloc(sizeof(double) * M), *w = malloc(sizeof(double) * M);
    double *beta = malloc(sizeof(double) * M), *v = malloc(sizeof(double) * M), *tmp = malloc(sizeof(double) * M);
    float *Mat = malloc(sizeof(float) * MM);
    double *Qr = malloc(sizeof(double) * MM), *QP = malloc(sizeof(double) * MM), *xis = malloc(sizeof(double) * MM);
    double *S = malloc(sizeof(double) * MM), *dsc = malloc(sizeof(double) * M);
    int rc = 0;

    memset(gamma_sums, 0, sizeof(double) * (size_t)K * M);
    memset(gs_present, 0, (size_t)K);
    if (L > 0) gs_present[kid[0]] = 1;                              /* hmm.cpp:53 */
    if (save_gamma) memset(gamma, 0, sizeof(double) * (size_t)(L + 1) * M);

    /* ---- forward, hmm.cpp:57-96 ---- */
    double ll = 0.;
    for (int m = 0; m < M; ++m) alpha_hat[m] = (float)pi[m];
    log_c[0] = 0.;
    for (int ell = 1; ell <= L; ++ell)
    {
        const int k = kid[ell - 1], sp = span[ell - 1];
        const double *b = E + (size_t)k * M;
        const float *ap = alpha_hat + (size_t)(ell - 1) * M;
        float *an = alpha_hat + (size_t)ell * M;
        gs_present[k] = 1;                                          /* hmm.cpp:69 */
        if (sp > 1 && eig_of_key[k] >= 0)
        {
            const int e = eig_of_key[k];
            const double *Pe = P + e * MM, *Pie = Pinv + e * MM, *de = d + (size_t)e * M;
            const double sc = scale[e];
            for (int i = 0; i < M; ++i)
            {
                double s = 0.;
                for (int j = 0; j < M; ++j) s += Pie[IDX(i, j, M)] * (double)ap[j];
                u[i] = pow(de[i] / sc, sp) * s;
            }
            for (int i = 0; i < M; ++i)
            {
                double s = 0.;
                for (int j = 0; j < M; ++j) s += Pe[IDX(i, j, M)] * u[j];
                a[i] = s;
            }
            double s = vsum(a, M);
            for (int i = 0; i < M; ++i) a[i] /= s;
            log_c[ell] = log(s) + sp * log(sc);
            for (int i = 0; i < M; ++i) an[i] = (float)a[i];
        }
        else
        {
            /* (B * T^T).pow(span) with span == 1 is exact; cast to float per element (hmm.cpp:85-86) */
            if (sp != 1)
            {
                /* the reference would compute a true matrix power here; only reachable when a span>1 row has no
                 * eigensystem, which fill_targets rules out (inference_manager.cpp:245-246) */
                rc = 1;
                goto done;
            }
            for (int i = 0; i < M; ++i)
                for (int kk = 0; kk < M; ++kk)
                    Mat[IDX(i, kk, M)] = (float)(b[i] * T[IDX(kk, i, M)]);
            for (int i = 0; i < M; ++i) an[i] = 0.f;
            for (int kk = 0; kk < M; ++kk)
            {
                const float x = 1.0f * ap[kk];
                for (int i = 0; i < M; ++i)
                {
                    const float prod = x * Mat[IDX(i, kk, M)];
                    an[i] = an[i] + prod;
                }
            }
            double s = (double)eigen_float_sum(an, M);
            log_c[ell] = log(s);
            const float sf = (float)s;
            for (int i = 0; i < M; ++i) an[i] /= sf;
        }
        for (int i = 0; i < M; ++i)
            if (an[i] < 1e-10f) an[i] = 1e-10f;
        ll += log_c[ell];
    }
    *loglik = ll;

    /* ---- backward, hmm.cpp:97-152 ---- */
    for (int i = 0; i < M; ++i) beta[i] = 1.;
    memset(xisum, 0, sizeof(double) * MM);
    for (int ell = L; ell > 0; --ell)
    {
        const int k = kid[ell - 1], sp = span[ell - 1];
        const double *b = E + (size_t)k * M;
        const float *ap = alpha_hat + (size_t)(ell - 1) * M;
        const float *ac = alpha_hat + (size_t)ell * M;
        if (beta_store) memcpy(beta_store + (size_t)ell * M, beta, sizeof(double) * M);
        if (sp > 1 && eig_of_key[k] >= 0)
        {
            const int e = eig_of_key[k];
            const double *Pe = P + e * MM, *Pie = Pinv + e * MM, *de = d + (size_t)e * M;
            const double sc = scale[e];
            const double log_p = log(sc) * (sp - 1);
            for (int i = 0; i < M; ++i) dsc[i] = de[i] / sc;
            oracle_span_q(M, sp, dsc, S);
            /* Q_r = (Pinv a)(beta^T P) o S */
            for (int i = 0; i < M; ++i)
            {
                double s = 0.;
                for (int j = 0; j < M; ++j) s += Pie[IDX(i, j, M)] * (double)ap[j];
                u[i] = s;
            }
            for (int j = 0; j < M; ++j)
            {
                double s = 0.;
                for (int i = 0; i < M; ++i) s += beta[i] * Pe[IDX(i, j, M)];
                w[j] = s;
            }
            for (int i = 0; i < M; ++i)
                for (int j = 0; j < M; ++j) Qr[IDX(i, j, M)] = u[i] * w[j] * S[IDX(i, j, M)];
            /* QP = Q_r * Pinv */
            for (int i = 0; i < M; ++i)
                for (int j = 0; j < M; ++j)
                {
                    double s = 0.;
                    for (int kk = 0; kk < M; ++kk) s += Qr[IDX(i, kk, M)] * Pie[IDX(kk, j, M)];
                    QP[IDX(i, j, M)] = s;
                }
            /* v = log|diag(P D QP)| - log_c + log_p */
            for (int i = 0; i < M; ++i)
            {
                double s = 0.;
                for (int j = 0; j < M; ++j) s += Pe[IDX(i, j, M)] * de[j] * QP[IDX(j, i, M)];
                v[i] = log(fabs(s)) - log_c[ell] + log(sc) * (sp - 1);
            }
            double vM = v[0];
            for (int i = 1; i < M; ++i) if (v[i] > vM) vM = v[i];
            double se = 0.;
            for (int i = 0; i < M; ++i) { v[i] -= vM; se += exp(v[i]); }
            const double log_C = log((double)sp) - vM - log(se);
            for (int i = 0; i < M; ++i) v[i] = exp(v[i] + vM + log_C);
            /* xis = exp(log|P QP B| - log_c + log_p + log_C) */
            for (int i = 0; i < M; ++i)
                for (int j = 0; j < M; ++j)
                {
                    double s = 0.;
                    for (int kk = 0; kk < M; ++kk) s += Pe[IDX(i, kk, M)] * QP[IDX(kk, j, M)];
                    xis[IDX(i, j, M)] = exp(log(fabs(s * b[j])) - log_c[ell] + log_p + log_C);
                }
            /* log_beta = log(Pinv^T (d~^span o (P^T beta))) + log_p + log_C + log(scale) */
            for (int j = 0; j < M; ++j) tmp[j] = pow(dsc[j], sp) * w[j];
            vM = -INFINITY;
            for (int i = 0; i < M; ++i)
            {
                double s = 0.;
                for (int j = 0; j < M; ++j) s += Pie[IDX(j, i, M)] * tmp[j];
                a[i] = log(s) + log_p + log_C + log(sc);
                if (a[i] > vM) vM = a[i];
            }
            for (int i = 0; i < M; ++i) beta[i] = exp(a[i] - vM);
        }
        else
        {
            if (sp != 1) { rc = 1; goto done; }
            for (int i = 0; i < M; ++i) v[i] = (double)ac[i] * beta[i];
            const double p = vsum(v, M);
            for (int i = 0; i < M; ++i) v[i] /= p;
            const double ec = exp(log_c[ell]);
            for (int i = 0; i < M; ++i)
                for (int j = 0; j < M; ++j)
                    xis[IDX(i, j, M)] = (double)ap[i] * beta[j] * b[j] / ec / p;
            for (int j = 0; j < M; ++j) tmp[j] = b[j] * beta[j];
            for (int i = 0; i < M; ++i)
            {
                double s = 0.;
                for (int j = 0; j < M; ++j) s += T[IDX(i, j, M)] * tmp[j];
                a[i] = s;
            }
            memcpy(beta, a, sizeof(double) * M);
        }
        for (size_t i = 0; i < MM; ++i) xisum[i] += xis[i];
        const double bs = vsum(beta, M);
        for (int i = 0; i < M; ++i) beta[i] /= bs;
        double *gs = gamma_sums + (size_t)k * M;
        for (int i = 0; i < M; ++i) gs[i] += v[i];
        if (save_gamma)
            memcpy(gamma + (size_t)ell * M, v, sizeof(double) * M);
    }
    if (beta_store) memcpy(beta_store, beta, sizeof(double) * M);
    for (int i = 0; i < M; ++i) gamma[i] = (double)alpha_hat[i] * beta[i];   /* gamma.col(0), hmm.cpp:150 */
    for (int i = 0; i < M; ++i)
        for (int j = 0; j < M; ++j)
        {
            double x = xisum[IDX(i, j, M)] * T[IDX(i, j, M)];
            xisum[IDX(i, j, M)] = x < 1e-20 ? 1e-20 : x;
        }
done:
    free(a); free(u); free(w); free(beta); free(v); free(tmp); free(Mat); free(Qr); free(QP); free(xis);
    free(S); free(dsc);
    return rc;
}

static double dcs(const double *x, size_t n)
{
    /* doubly_compensated_summation, common.h:27-46 */
    if (n == 0) return 0.0;
    double s = x[0], c = 0.0;
    for (size_t i = 1; i < n; ++i)
    {
        double y = c + x[i];
        double u = x[i] - (y - c);
        double t = y + s;
        double v = y - (t - s);
        double z = u + v;
        s = t + z;
        c = z - (s - t);
    }
    return s;
}

/* HMM::Q values.  nb_pos[K] = 1 if key.nb() > 0.  gamma0 [M], gamma_sums [K x M] with gs_present[K], keys visited
 * in index order (the caller sorts key ids in the reference's std::map order). */
void oracle_q(int M, int K, const double *E, const double *pi, const double *T,
              const double *gamma0, const double *gamma_sums, const unsigned char *gs_present,
              const unsigned char *nb_pos, const double *xisum, double *q)
{
    double q0 = 0.;
    for (int m = 0; m < M; ++m) q0 += log(pi[m]) * gamma0[m];
    q[0] = q0;
    double *buf[2];
    size_t cnt[2] = {0, 0};
    buf[0] = malloc(sizeof(double) * (size_t)K * M + 8);
    buf[1] = malloc(sizeof(double) * (size_t)K * M + 8);
    int bad[2] = {0, 0};
    for (int k = 0; k < K; ++k)
    {
        if (!gs_present[k]) continue;
        const int i = nb_pos[k] ? 1 : 0;
        const double *e = E + (size_t)k * M;
        double mn = e[0];
        for (int m = 1; m < M; ++m) if (e[m] < mn) mn = e[m];
        if (mn <= 0.0) { bad[i] = 1; break; }     /* hmm.cpp:169-175 (UB in the reference; -inf here) */
        for (int m = 0; m < M; ++m) buf[i][cnt[i]++] = log(e[m]) * gamma_sums[(size_t)k * M + m];
    }
    q[1] = bad[0] ? -INFINITY : dcs(buf[0], cnt[0]);
    q[2] = bad[1] ? -INFINITY : dcs(buf[1], cnt[1]);
    free(buf[0]); free(buf[1]);
    /* prod.data() is column-major in the reference (hmm.cpp:183-184) */
    double *es = malloc(sizeof(double) * (size_t)M * M);
    for (int j = 0; j < M; ++j)
        for (int i = 0; i < M; ++i)
            es[(size_t)j * M + i] = log(T[IDX(i, j, M)]) * xisum[IDX(i, j, M)];
    q[3] = dcs(es, (size_t)M * M);
    free(es);
}
