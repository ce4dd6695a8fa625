/* Davies (1980), Algorithm AS 155: distribution of a linear combination of
 * chi-squared variables -- CPU oracle (TEST INFRASTRUCTURE ONLY).
 *
 * The reference reaches this through chiscore.davies_pvalue -> chi2comb
 * (C library "chi2comb_cdf", a re-implementation of Davies' qfc); call site
 * /root/reference cellregmap/_cellregmap.py:333,435.  chi2comb is absent from
 * this image, so this file restates the published algorithm (Applied
 * Statistics 29:323-333) in plain C.  PARITY UNPINNED: the reference holds no
 * golden Davies p-value; tests pin this file against a numerical Imhof
 * integral and closed forms instead (tests/test_oracle_davies.py).
 *
 *   P[ sum_j lb[j] * chi2(n[j], nc[j]) + sigma * N(0,1)  <  c ]
 *
 * ifault: 0 ok; 1 required accuracy not reached within lim terms; 2 round-off
 * possibly significant; 3 invalid parameters; 4 unable to locate integration
 * parameters (evaluation counter exceeded lim).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define CRM_PI 3.14159265358979323846
#define CRM_LN28 0.08664339756999316 /* log(2)/8 */

typedef struct {
    const double *lb, *nc;
    const int *n;
    int r, lim, count, ordered, fail, overflow;
    int *th;
    double sigsq, lmax, lmin, mean, c, intl, ersm;
} qf_state;

static double exp_guard(double x) { return x < -50.0 ? 0.0 : exp(x); }

/* log(1+x) when first, else log(1+x) - x; series for small |x| */
static double log1p_variant(double x, int first)
{
    if (fabs(x) > 0.1) return first ? log(1.0 + x) : (log(1.0 + x) - x);
    double y = x / (2.0 + x);
    double term = 2.0 * y * y * y;
    double k = 3.0;
    double s = (first ? 2.0 : -x) * y;
    y = y * y;
    double s1 = s + term / k;
    while (s1 != s) {
        k += 2.0;
        term *= y;
        s = s1;
        s1 = s + term / k;
    }
    return s;
}

static void tick(qf_state *q)
{
    q->count++;
    if (q->count > q->lim) q->overflow = 1;
}

/* insertion order of |lb| descending into th */
static void sort_by_abs(qf_state *q)
{
    for (int j = 0; j < q->r; j++) {
        double lj = fabs(q->lb[j]);
        int k = j - 1;
        while (k >= 0 && lj > fabs(q->lb[q->th[k]])) {
            q->th[k + 1] = q->th[k];
            k--;
        }
        q->th[k + 1] = j;
    }
    q->ordered = 1;
}

/* bound on tail probability via the mgf; cut-off point returned in *cx */
static double tail_bound(qf_state *q, double u, double *cx)
{
    tick(q);
    double xconst = u * q->sigsq;
    double sum1 = u * xconst;
    u = 2.0 * u;
    for (int j = q->r - 1; j >= 0; j--) {
        double nj = q->n[j], lj = q->lb[j], ncj = q->nc[j];
        double x = u * lj, y = 1.0 - x;
        xconst += lj * (ncj / y + nj) / y;
        double xy = x / y;
        sum1 += ncj * xy * xy + nj * (x * x / y + log1p_variant(-x, 0));
    }
    *cx = xconst;
    return exp_guard(-0.5 * sum1);
}

/* find cut-off so that P(qf > ctff) < accx (upn>0) or P(qf < ctff) < accx */
static double cutoff(qf_state *q, double accx, double *upn)
{
    double u2 = *upn, u1 = 0.0, c1 = q->mean, c2 = 0.0, xconst;
    double rb = 2.0 * ((u2 > 0.0) ? q->lmax : q->lmin);
    double u = u2 / (1.0 + u2 * rb);
    while (!q->overflow && tail_bound(q, u, &c2) > accx) {
        u1 = u2;
        c1 = c2;
        u2 = 2.0 * u2;
        u = u2 / (1.0 + u2 * rb);
    }
    u = (c1 - q->mean) / (c2 - q->mean);
    while (!q->overflow && u < 0.9) {
        u = (u1 + u2) / 2.0;
        if (tail_bound(q, u / (1.0 + u * rb), &xconst) > accx) {
            u1 = u;
            c1 = xconst;
        } else {
            u2 = u;
            c2 = xconst;
        }
        u = (c1 - q->mean) / (c2 - q->mean);
    }
    *upn = u2;
    return c2;
}

/* bound on the integration error due to truncation at u */
static double trunc_bound(qf_state *q, double u, double tausq)
{
    tick(q);
    double sum1 = 0.0, prod2 = 0.0, prod3 = 0.0;
    int s = 0;
    double sum2 = (q->sigsq + tausq) * u * u;
    double prod1 = 2.0 * sum2;
    u = 2.0 * u;
    for (int j = 0; j < q->r; j++) {
        double lj = q->lb[j], ncj = q->nc[j];
        int nj = q->n[j];
        double x = (u * lj) * (u * lj);
        sum1 += ncj * x / (1.0 + x);
        if (x > 1.0) {
            prod2 += nj * log(x);
            prod3 += nj * log1p_variant(x, 1);
            s += nj;
        } else {
            prod1 += nj * log1p_variant(x, 1);
        }
    }
    sum1 *= 0.5;
    prod2 += prod1;
    prod3 += prod1;
    double x = exp_guard(-sum1 - 0.25 * prod2) / CRM_PI;
    double y = exp_guard(-sum1 - 0.25 * prod3) / CRM_PI;
    double err1 = (s == 0) ? 1.0 : x * 2.0 / s;
    double err2 = (prod3 > 1.0) ? 2.5 * y : 1.0;
    if (err2 < err1) err1 = err2;
    x = 0.5 * sum2;
    err2 = (x <= y) ? 1.0 : y / x;
    return (err1 < err2) ? err1 : err2;
}

/* find u with trunc_bound(u) < accx and trunc_bound(u/1.2) > accx */
static void find_trunc_point(qf_state *q, double *utx, double accx)
{
    static const double divis[4] = {2.0, 1.4, 1.2, 1.1};
    double ut = *utx, u = ut / 4.0;
    if (trunc_bound(q, u, 0.0) > accx) {
        for (u = ut; !q->overflow && trunc_bound(q, u, 0.0) > accx; u = ut) ut *= 4.0;
    } else {
        ut = u;
        for (u = u / 4.0; !q->overflow && trunc_bound(q, u, 0.0) <= accx; u = u / 4.0) ut = u;
    }
    for (int i = 0; i < 4; i++) {
        u = ut / divis[i];
        if (trunc_bound(q, u, 0.0) <= accx) ut = u;
    }
    *utx = ut;
}

/* trapezoid sum with nterm+1 terms at step interv; when !mainx the integrand
 * is multiplied by 1 - exp(-0.5 tausq u^2) */
static void integrate(qf_state *q, int nterm, double interv, double tausq, int mainx)
{
    double inpi = interv / CRM_PI;
    for (int k = nterm; k >= 0; k--) {
        double u = (k + 0.5) * interv;
        double sum1 = -2.0 * u * q->c;
        double sum2 = fabs(sum1);
        double sum3 = -0.5 * q->sigsq * u * u;
        for (int j = q->r - 1; j >= 0; j--) {
            int nj = q->n[j];
            double x = 2.0 * q->lb[j] * u;
            double y = x * x;
            sum3 -= 0.25 * nj * log1p_variant(y, 1);
            y = q->nc[j] * x / (1.0 + y);
            double z = nj * atan(x) + y;
            sum1 += z;
            sum2 += fabs(z);
            sum3 -= 0.5 * x * y;
        }
        double x = inpi * exp_guard(sum3) / u;
        if (!mainx) x *= (1.0 - exp_guard(-0.5 * tausq * u * u));
        sum1 = sin(0.5 * sum1) * x;
        sum2 = 0.5 * sum2 * x;
        q->intl += sum1;
        q->ersm += sum2;
    }
}

/* coefficient of tausq in the error when the convergence factor
 * exp(-0.5 tausq u^2) is used and the df is evaluated at x */
static double conv_coef(qf_state *q, double x)
{
    tick(q);
    if (!q->ordered) sort_by_abs(q);
    double axl = fabs(x), sxl = (x > 0.0) ? 1.0 : -1.0, sum1 = 0.0;
    for (int j = q->r - 1; j >= 0; j--) {
        int t = q->th[j];
        if (q->lb[t] * sxl > 0.0) {
            double lj = fabs(q->lb[t]);
            double axl1 = axl - lj * (q->n[t] + q->nc[t]);
            double axl2 = lj / CRM_LN28;
            if (axl1 > axl2) {
                axl = axl1;
            } else {
                if (axl > axl2) axl = axl2;
                sum1 = (axl - axl1) / lj;
                for (int k = j - 1; k >= 0; k--) sum1 += (q->n[q->th[k]] + q->nc[q->th[k]]);
                break;
            }
        }
    }
    if (sum1 > 100.0) {
        q->fail = 1;
        return 1.0;
    }
    return pow(2.0, sum1 / 4.0) / (CRM_PI * axl * axl);
}

/* trace[7]: 0 abs-sum, 1 total terms, 2 integrations, 3 main interval,
 * 4 truncation point, 5 sd of convergence factor, 6 counter */
int crm_oracle_qfc(const double *lb, const double *nc, const int *n, int r, double sigma,
                   double c, int lim, double acc, double *trace, int *ifault, double *res)
{
    static const int rats[4] = {1, 2, 4, 8};
    qf_state q;
    memset(&q, 0, sizeof q);
    q.lb = lb; q.nc = nc; q.n = n; q.r = r; q.lim = lim; q.c = c;
    for (int j = 0; j < 7; j++) trace[j] = 0.0;
    *ifault = 0;
    double qfval = -1.0, acc1 = acc, xlim = (double)lim;
    q.th = (int *)malloc((r > 0 ? r : 1) * sizeof(int));
    if (!q.th) { *ifault = 5; *res = qfval; return 5; }

    q.sigsq = sigma * sigma;
    double sd = q.sigsq;
    for (int j = 0; j < r; j++) {
        int nj = n[j];
        double lj = lb[j], ncj = nc[j];
        if (nj < 0 || ncj < 0.0) { *ifault = 3; goto done; }
        sd += lj * lj * (2 * nj + 4.0 * ncj);
        q.mean += lj * (nj + ncj);
        if (q.lmax < lj) q.lmax = lj;
        else if (q.lmin > lj) q.lmin = lj;
    }
    if (sd == 0.0) { qfval = (c > 0.0) ? 1.0 : 0.0; goto done; }
    if (q.lmin == 0.0 && q.lmax == 0.0 && sigma == 0.0) { *ifault = 3; goto done; }
    sd = sqrt(sd);
    double almx = (q.lmax < -q.lmin) ? -q.lmin : q.lmax;

    double utx = 16.0 / sd, up = 4.5 / sd, un = -up, tausq, intv, d1, d2, xnt, xntm;
    find_trunc_point(&q, &utx, 0.5 * acc1);
    if (q.overflow) { *ifault = 4; goto done; }
    if (c != 0.0 && almx > 0.07 * sd) {
        tausq = 0.25 * acc1 / conv_coef(&q, c);
        if (q.fail) {
            q.fail = 0;
        } else if (trunc_bound(&q, utx, tausq) < 0.2 * acc1) {
            q.sigsq += tausq;
            find_trunc_point(&q, &utx, 0.25 * acc1);
            trace[5] = sqrt(tausq);
        }
        if (q.overflow) { *ifault = 4; goto done; }
    }
    trace[4] = utx;
    acc1 *= 0.5;

    for (;;) {
        d1 = cutoff(&q, acc1, &up) - c;
        if (q.overflow) { *ifault = 4; goto done; }
        if (d1 < 0.0) { qfval = 1.0; goto done; }
        d2 = c - cutoff(&q, acc1, &un);
        if (q.overflow) { *ifault = 4; goto done; }
        if (d2 < 0.0) { qfval = 0.0; goto done; }
        intv = 2.0 * CRM_PI / ((d1 > d2) ? d1 : d2);
        xnt = utx / intv;
        xntm = 3.0 / sqrt(acc1);
        if (xnt <= xntm * 1.5) break;
        /* auxiliary integration */
        if (xntm > xlim) { *ifault = 1; goto done; }
        int ntm = (int)floor(xntm + 0.5);
        double intv1 = utx / ntm;
        double x = 2.0 * CRM_PI / intv1;
        if (x <= fabs(c)) break;
        tausq = 0.33 * acc1 / (1.1 * (conv_coef(&q, c - x) + conv_coef(&q, c + x)));
        if (q.overflow) { *ifault = 4; goto done; }
        if (q.fail) break;
        acc1 *= 0.67;
        integrate(&q, ntm, intv1, tausq, 0);
        xlim -= xntm;
        q.sigsq += tausq;
        trace[2] += 1.0;
        trace[1] += ntm + 1;
        find_trunc_point(&q, &utx, 0.25 * acc1);
        if (q.overflow) { *ifault = 4; goto done; }
        acc1 *= 0.75;
    }

    trace[3] = intv;
    if (xnt > xlim) { *ifault = 1; goto done; }
    {
        int nt = (int)floor(xnt + 0.5);
        integrate(&q, nt, intv, 0.0, 1);
        trace[2] += 1.0;
        trace[1] += nt + 1;
        qfval = 0.5 - q.intl;
        trace[0] = q.ersm;
        double upv = q.ersm, x = upv + acc / 10.0;
        for (int j = 0; j < 4; j++)
            if (rats[j] * x == rats[j] * upv) *ifault = 2;
    }

done:
    free(q.th);
    trace[6] = (double)q.count;
    *res = qfval;
    return *ifault;
}
