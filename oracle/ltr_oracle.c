/*
 * oracle/ltr_oracle.c -- TEST INFRASTRUCTURE ONLY.  NOT part of the product.
 *
 * A plain-C, double-precision, scalar CPU restatement of the pytorchltr
 * ranking-loss / ranking-metric hot path.  It exists so that the HIP kernels
 * in pytorchltr_amd/csrc can be checked against something that follows the
 * reference's arithmetic literally.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it; the product path never does.
 *
 * Every function cites the reference file:line (relative to the upstream
 * rjagerman/pytorchltr tree, v0.2.1) whose behaviour it restates.  The code is
 * written from the reference's *semantics* (per-query loops over document
 * pairs); the reference itself is tensor code that materialises (B,L,L,2)
 * pair tensors and has no loops.
 *
 * Parity is PINNED: tests/test_oracle_golden.py checks this file against
 *   (a) every literal known-answer case of the reference's own unit tests
 *       (tests/loss/test_pairwise_additive.py, tests/loss/test_pairwise_lambda.py,
 *        tests/evaluation/test_dcg.py, tests/evaluation/test_arp.py, docs examples),
 *   (b) vectors produced by importing the real reference in the build
 *       container (tests/golden/generate_golden.py -> tests/golden/*.npz),
 *       including autograd gradients for all seven losses.
 *
 * Deliberate, documented deviation: the reference breaks
 * score ties with a random permutation drawn from torch's global RNG
 * (utils/tensor_operations.py:43-45).  This oracle -- and the HIP path --
 * use the deterministic rule "masked score descending, then document index
 * ascending"; the padded tail therefore comes out in index order.  Results
 * agree with the reference on every row whose first n[b] scores are tie-free.
 *
 * Build:  gcc -O2 -fPIC -shared -o oracle/_build/libltr_oracle.so oracle/ltr_oracle.c -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum {
    LTR_HINGE = 0,      /* PairwiseHingeLoss      loss/pairwise_additive.py:93-113  */
    LTR_DCG_HINGE = 1,  /* PairwiseDCGHingeLoss   loss/pairwise_additive.py:116-133 */
    LTR_LOGISTIC = 2,   /* PairwiseLogisticLoss   loss/pairwise_additive.py:136-163 */
    LTR_ARP1 = 3,       /* LambdaARPLoss1         loss/pairwise_lambda.py:95-117    */
    LTR_ARP2 = 4,       /* LambdaARPLoss2         loss/pairwise_lambda.py:120-140   */
    LTR_NDCG1 = 5,      /* LambdaNDCGLoss1        loss/pairwise_lambda.py:143-173   */
    LTR_NDCG2 = 6       /* LambdaNDCGLoss2        loss/pairwise_lambda.py:176-218   */
};

#define LN2 0.69314718055994530942

/* n[b] as the reference's masks see it: `arange >= n` (tensor_operations.py:25)
 * and `n_grid <= range_grid` (pairwise_additive.py:75-81) keep exactly the
 * documents with index < n; n > L keeps all L, n <= 0 keeps none. */
static int clamp_n(int64_t n, int L)
{
    if (n < 0) return 0;
    if (n > (int64_t)L) return L;
    return (int)n;
}

/* mask_padded_values, utils/tensor_operations.py:6-26:
 * out[b,j] = j >= n[b] ? mask_value : xs[b,j]. */
int oracle_mask_padded_values(const double *xs, const int64_t *n, int B, int L,
                              double mask_value, double *out)
{
    for (int b = 0; b < B; ++b) {
        int nb = clamp_n(n[b], L);
        for (int j = 0; j < L; ++j)
            out[(size_t)b * L + j] = (j >= nb) ? mask_value : xs[(size_t)b * L + j];
    }
    return 0;
}

/* batch_pairs, utils/tensor_operations.py:94-119:
 * p[b,i,j,0] = x[b,i]; p[b,i,j,1] = x[b,j]. */
int oracle_batch_pairs(const double *x, int B, int L, double *out)
{
    for (int b = 0; b < B; ++b)
        for (int i = 0; i < L; ++i)
            for (int j = 0; j < L; ++j) {
                size_t o = ((((size_t)b * L) + i) * L + j) * 2;
                out[o + 0] = x[(size_t)b * L + i];
                out[o + 1] = x[(size_t)b * L + j];
            }
    return 0;
}

/* One row of rank_by_score, utils/tensor_operations.py:48-64 (which is
 * mask_padded_values(-inf) :6-26 followed by a descending argsort :29-45).
 * ranking[r] = index of the document at rank r.  Tie rule: see file header. */
/* Optional tie priorities (oracle_set_tie): the reference's tiebreak_argsort,
 * utils/tensor_operations.py:29-45, lets ONE random permutation shared by all rows decide the
 * order of equal scores; with priorities set, of two REAL documents with equal score the one
 * with the smaller priority ranks first (NULL: index order, the deterministic default).  The
 * padded tail stays in index order either way. */
static const int32_t *g_tie = NULL;
void oracle_set_tie(const int32_t *tie) { g_tie = tie; }

static int tie_before(int m, int j, int nb)
{
    if (g_tie && m < nb && j < nb) return g_tie[m] < g_tie[j];
    return m < j;
}

static void rank_row(const double *s, int nb, int L, int64_t *ranking)
{
    for (int j = 0; j < L; ++j) {
        double kj = (j < nb) ? s[j] : -INFINITY;
        int r = 0;
        for (int m = 0; m < L; ++m) {
            double km = (m < nb) ? s[m] : -INFINITY;
            if (km > kj || (km == kj && tie_before(m, j, nb))) ++r;
        }
        ranking[r] = j;
    }
}

int oracle_rank_by_score(const double *scores, const int64_t *n, int B, int L,
                         int64_t *ranking)
{
    for (int b = 0; b < B; ++b)
        rank_row(scores + (size_t)b * L, clamp_n(n[b], L), L, ranking + (size_t)b * L);
    return 0;
}

/* _max_dcg, loss/pairwise_lambda.py:231-241, for one query.  `rel` is the
 * label row it is handed (the LambdaLoss hands it the score-sorted labels,
 * pairwise_lambda.py:225); sort labels descending over the first nb, zero the
 * rest (:238), gains 2^g-1 (:239-240), discounts log2(2+r) (:236), sum. */
static double max_dcg_row(const double *rel, int nb, int L, int64_t *tmp_rank)
{
    rank_row(rel, nb, L, tmp_rank);
    double acc = 0.0;
    for (int r = 0; r < L; ++r) {
        double g = (r < nb) ? rel[tmp_rank[r]] : 0.0;
        acc += (pow(2.0, g) - 1.0) / log2(2.0 + (double)r);
    }
    return acc;
}

/* Additive losses for one query: _PairwiseAdditiveLoss.forward,
 * loss/pairwise_additive.py:51-90.  Pairs (i,j) with max(i,j) >= n are zeroed
 * (:75-81), the rest summed (:45,:84).  Returns the un-modified pair sum and
 * accumulates d(sum)/d(s) into g[0..L). */
static double additive_row(int kind, double sigma, const double *s, const double *y,
                           int nb, double *g)
{
    double acc = 0.0;
    for (int i = 0; i < nb; ++i)
        for (int j = 0; j < nb; ++j) {
            if (!(y[i] - y[j] > 0.0)) continue;     /* loss[rel_pair_diffs <= 0] = 0 (:111,:162) */
            double d = s[i] - s[j];
            if (kind == LTR_LOGISTIC) {
                /* log2(1 + exp(-sigma*d))  (:161) */
                double e = exp(-sigma * d);
                acc += log2(1.0 + e);
                double dd = -sigma * e / (1.0 + e) / LN2;
                if (g) { g[i] += dd; g[j] -= dd; }
            } else {
                /* hinge: 1 - d, zeroed where < 0 (strict: at exactly the margin
                 * the term is 0 but its gradient still flows, :110-112) */
                double t = 1.0 - d;
                if (t < 0.0) continue;
                acc += t;
                if (g) { g[i] -= 1.0; g[j] += 1.0; }
            }
        }
    return acc;
}

/* LambdaLoss.forward for one query, loss/pairwise_lambda.py:50-92:
 * rank by score (:67), gather scores+labels into rank order (:68-70), evaluate
 * the per-pair term on rank positions (i,j) (:77), zero pairs touching
 * positions >= n (:80-86), sum (:89).  Gradients flow back through the gather
 * only (the ranking itself is a constant), so position-i gradients are
 * scattered to document ranking[i]. */
static double lambda_row(int kind, double sigma, const double *s, const double *y,
                         int nb, int L, double *g, int64_t *ranking, int64_t *tmp_rank,
                         double *ss, double *ys, double *gs)
{
    rank_row(s, nb, L, ranking);
    for (int r = 0; r < L; ++r) { ss[r] = s[ranking[r]]; ys[r] = y[ranking[r]]; gs[r] = 0.0; }

    double maxdcg = 1.0;
    if (kind == LTR_NDCG1 || kind == LTR_NDCG2) {
        /* _ndcg_gains, pairwise_lambda.py:221-228 */
        maxdcg = max_dcg_row(ys, nb, L, tmp_rank);
        if (maxdcg == 0.0) maxdcg = 1.0;           /* :227 */
    }

    double acc = 0.0;
    for (int i = 0; i < nb; ++i)
        for (int j = 0; j < nb; ++j) {
            double d = ss[i] - ss[j];
            double w;
            switch (kind) {
            case LTR_ARP1:                          /* sigmoid ** y_i  (:117) */
                w = ys[i];
                break;
            case LTR_ARP2:                          /* rel_diffs * log2(1+e^-sd), zero where rel_diffs<=0 (:136-140) */
                if (!(ys[i] - ys[j] > 0.0)) continue;
                w = ys[i] - ys[j];
                break;
            case LTR_NDCG1: {                       /* exponent = G_i / log2(2+i)  (:165-173) */
                double Gi = (pow(2.0, ys[i]) - 1.0) / maxdcg;
                w = Gi / log2(2.0 + (double)i);
                break;
            }
            default: {                              /* NDCG2 (:198-218) */
                if (!(ys[i] - ys[j] > 0.0)) continue;
                double Gi = (pow(2.0, ys[i]) - 1.0) / maxdcg;
                double Gj = (pow(2.0, ys[j]) - 1.0) / maxdcg;
                int k = abs(i - j);
                /* D(k) = log2(2+k): follow the code (:206-211), not the docstring */
                double delta = fabs(1.0 / log2(2.0 + (double)k) - 1.0 / log2(3.0 + (double)k));
                w = delta * fabs(Gi - Gj);
                break;
            }
            }
            /* every Lambda term is  -log2( sigmoid(sigma*d) ** w )  (ARP2 is the
             * same quantity written as w*log2(1+e^{-sigma d})) */
            double sig = 1.0 / (1.0 + exp(-sigma * d));
            if (kind == LTR_ARP2)
                acc += w * log2(1.0 + exp(-sigma * d));
            else
                acc += -log2(pow(sig, w));
            double dd = -w * sigma * (1.0 - sig) / LN2;   /* d term / d d */
            gs[i] += dd;
            gs[j] -= dd;
        }
    if (g)
        for (int r = 0; r < nb; ++r) g[ranking[r]] += gs[r];
    return acc;
}

/* All seven losses, forward + d loss[b] / d scores[b,:].
 * scores, rel: (B,L) row-major doubles; n: (B); loss: (B); dscores: (B,L) or NULL. */
int oracle_pairwise_loss(int kind, double sigma, const double *scores, const double *rel,
                         const int64_t *n, int B, int L, double *loss, double *dscores)
{
    if (kind < LTR_HINGE || kind > LTR_NDCG2 || B < 0 || L < 0) return -1;
    size_t Ls = (size_t)(L > 0 ? L : 1);
    int64_t *ranking = (int64_t *)malloc(sizeof(int64_t) * Ls);
    int64_t *tmp_rank = (int64_t *)malloc(sizeof(int64_t) * Ls);
    double *buf = (double *)malloc(sizeof(double) * Ls * 4);
    if (!ranking || !tmp_rank || !buf) { free(ranking); free(tmp_rank); free(buf); return -2; }
    double *ss = buf, *ys = buf + Ls, *gs = buf + 2 * Ls, *grow = buf + 3 * Ls;

    for (int b = 0; b < B; ++b) {
        const double *s = scores + (size_t)b * L;
        const double *y = rel + (size_t)b * L;
        int nb = clamp_n(n[b], L);
        for (int j = 0; j < L; ++j) grow[j] = 0.0;
        double acc;
        if (kind <= LTR_LOGISTIC)
            acc = additive_row(kind, sigma, s, y, nb, grow);
        else
            acc = lambda_row(kind, sigma, s, y, nb, L, grow, ranking, tmp_rank, ss, ys, gs);
        double scale = 1.0;
        if (kind == LTR_DCG_HINGE) {
            /* _loss_modifier: -1/ln(2+H)  (pairwise_additive.py:132-133);
             * d/dH = 1 / ((2+H) ln^2(2+H)) */
            double lg = log(2.0 + acc);
            scale = 1.0 / ((2.0 + acc) * lg * lg);
            acc = -1.0 / lg;
        }
        loss[b] = acc;
        if (dscores)
            for (int j = 0; j < L; ++j) dscores[(size_t)b * L + j] = scale * grow[j];
    }
    free(ranking); free(tmp_rank); free(buf);
    return 0;
}

/* dcg / ndcg, evaluation/dcg.py:41-99 and :8-38.
 * k <= 0  -> full curve, out is (B,L);  k > 0 -> out is (B), column min(k,L)-1 (:97-98).
 * Quirk kept from the reference: padded documents are NOT masked here, their
 * labels contribute at ranks >= n (harmless when collate zero-pads). */
static void dcg_row(const double *key, const double *rel, int nb, int L, int use_exp,
                    int64_t *ranking, double *curve)
{
    rank_row(key, nb, L, ranking);                  /* :85 */
    double acc = 0.0;
    for (int r = 0; r < L; ++r) {
        double g = rel[ranking[r]];
        if (use_exp) g = pow(2.0, g) - 1.0;         /* :91-92 */
        acc += g / log2((double)r + 2.0);           /* :93-94 */
        curve[r] = acc;
    }
}

int oracle_dcg(const double *scores, const double *rel, const int64_t *n, int B, int L,
               int k, int use_exp, int normalize, double *out)
{
    if (B < 0 || L <= 0) return -1;
    int64_t *ranking = (int64_t *)malloc(sizeof(int64_t) * (size_t)L);
    double *curve = (double *)malloc(sizeof(double) * (size_t)L * 2);
    if (!ranking || !curve) { free(ranking); free(curve); return -2; }
    double *icurve = curve + L;
    int col = (k > 0) ? ((k < L ? k : L) - 1) : -1;
    for (int b = 0; b < B; ++b) {
        const double *s = scores + (size_t)b * L;
        const double *y = rel + (size_t)b * L;
        int nb = clamp_n(n[b], L);
        dcg_row(s, y, nb, L, use_exp, ranking, curve);
        if (normalize) dcg_row(y, y, nb, L, use_exp, ranking, icurve);   /* dcg.py:36 */
        if (col >= 0) {
            double v = curve[col];
            if (normalize) { double id = icurve[col]; if (id == 0.0) id = 1.0; v /= id; }  /* :37-38 */
            out[b] = v;
        } else {
            for (int r = 0; r < L; ++r) {
                double v = curve[r];
                if (normalize) { double id = icurve[r]; if (id == 0.0) id = 1.0; v /= id; }
                out[(size_t)b * L + r] = v;
            }
        }
    }
    free(ranking); free(curve);
    return 0;
}

/* arp, evaluation/arp.py:7-42: sort, gather labels, zero ranks >= n (:38),
 * sum((r+1)*rel_r) / sum(rel_r) with a 0 -> 1 guard on the denominator (:41). */
int oracle_arp(const double *scores, const double *rel, const int64_t *n, int B, int L,
               double *out)
{
    if (B < 0 || L <= 0) return -1;
    int64_t *ranking = (int64_t *)malloc(sizeof(int64_t) * (size_t)L);
    if (!ranking) return -2;
    for (int b = 0; b < B; ++b) {
        const double *s = scores + (size_t)b * L;
        const double *y = rel + (size_t)b * L;
        int nb = clamp_n(n[b], L);
        rank_row(s, nb, L, ranking);
        double srp = 0.0, nrp = 0.0;
        for (int r = 0; r < nb; ++r) {
            double v = y[ranking[r]];
            srp += (double)(r + 1) * v;
            nrp += v;
        }
        if (nrp == 0.0) nrp = 1.0;
        out[b] = srp / nrp;
    }
    free(ranking);
    return 0;
}

/* Linear scorer + loss, the caller of the path in every reference workflow
 * (examples/01-basic-usage.py:70-75: loss_fn(model(xs), ys, n).mean().backward()
 * with model = torch.nn.Linear(F, 1)).  scores = X.W + b; returns per-query loss
 * and d(sum_b gout[b]*loss[b]) / d{W, b}.  X: (B,L,F); W: (F); gout: (B). */
int oracle_linear_pairwise(int kind, double sigma, const double *X, const double *W, double bias,
                           const double *rel, const int64_t *n, const double *gout,
                           int B, int L, int F, double *loss, double *scores_out,
                           double *dW, double *db)
{
    double *scores = (double *)malloc(sizeof(double) * (size_t)B * L);
    double *ds = (double *)malloc(sizeof(double) * (size_t)B * L);
    if (!scores || !ds) { free(scores); free(ds); return -2; }
    for (size_t r = 0; r < (size_t)B * L; ++r) {
        double acc = bias;
        for (int f = 0; f < F; ++f) acc += X[r * F + f] * W[f];
        scores[r] = acc;
    }
    int rc = oracle_pairwise_loss(kind, sigma, scores, rel, n, B, L, loss, ds);
    if (rc == 0) {
        for (int f = 0; f < F; ++f) dW[f] = 0.0;
        *db = 0.0;
        for (int b = 0; b < B; ++b)
            for (int l = 0; l < L; ++l) {
                size_t r = (size_t)b * L + l;
                double gr = gout[b] * ds[r];
                if (gr == 0.0) continue;
                *db += gr;
                for (int f = 0; f < F; ++f) dW[f] += gr * X[r * F + f];
            }
        if (scores_out) memcpy(scores_out, scores, sizeof(double) * (size_t)B * L);
    }
    free(scores); free(ds);
    return rc;
}

/* ---------------------------------------------------------------------------------------
 * ReLU MLP scorer (F -> H1 -> H2 -> 1) + pairwise loss + backward: the documented training
 * step `loss_fn(model(xs), ys, n)` weighted by gout[b], with `model` the feed-forward network
 * of docs/source/getting-started.rst:40-50 (l1, relu, l2, relu, l3).  Weights in torch.nn.Linear
 * layout: W1 (H1,F), W2 (H2,H1), W3 (1,H2).  grads = [dW1 | db1 | dW2 | db2 | dW3 | db3].
 * relu'(0) = 0 as in torch.
 * --------------------------------------------------------------------------------------- */
int oracle_mlp_pairwise(int kind, double sigma, const double *X, const double *W1,
                        const double *b1, const double *W2, const double *b2, const double *W3,
                        double b3, const double *rel, const int64_t *n, const double *gout,
                        int B, int L, int F, int H1, int H2, double *loss, double *scores_out,
                        double *grads)
{
    const size_t R = (size_t)B * L;
    double *scores = (double *)malloc(sizeof(double) * R);
    double *ds = (double *)malloc(sizeof(double) * R);
    double *h1 = (double *)malloc(sizeof(double) * (size_t)H1);
    double *h2 = (double *)malloc(sizeof(double) * (size_t)H2);
    double *d1 = (double *)malloc(sizeof(double) * (size_t)H1);
    double *d2 = (double *)malloc(sizeof(double) * (size_t)H2);
    if (!scores || !ds || !h1 || !h2 || !d1 || !d2) {
        free(scores); free(ds); free(h1); free(h2); free(d1); free(d2);
        return -2;
    }
    for (size_t r = 0; r < R; ++r) {
        const double *x = X + r * F;
        for (int j = 0; j < H1; ++j) {
            double a = b1[j];
            for (int f = 0; f < F; ++f) a += W1[(size_t)j * F + f] * x[f];
            h1[j] = a > 0.0 ? a : 0.0;
        }
        double s = b3;
        for (int k = 0; k < H2; ++k) {
            double a = b2[k];
            for (int j = 0; j < H1; ++j) a += W2[(size_t)k * H1 + j] * h1[j];
            s += W3[k] * (a > 0.0 ? a : 0.0);
        }
        scores[r] = s;
    }
    int rc = oracle_pairwise_loss(kind, sigma, scores, rel, n, B, L, loss, ds);
    if (rc == 0) {
        double *dW1 = grads, *db1 = dW1 + (size_t)H1 * F, *dW2 = db1 + H1;
        double *db2 = dW2 + (size_t)H2 * H1, *dW3 = db2 + H2, *db3 = dW3 + H2;
        const size_t P = (size_t)H1 * F + H1 + (size_t)H2 * H1 + 2 * (size_t)H2 + 1;
        for (size_t i = 0; i < P; ++i) grads[i] = 0.0;
        for (int b = 0; b < B; ++b)
            for (int l = 0; l < L; ++l) {
                const size_t r = (size_t)b * L + l;
                const double gr = gout[b] * ds[r];
                if (gr == 0.0) continue;
                const double *x = X + r * F;
                for (int j = 0; j < H1; ++j) {              /* recompute the activations */
                    double a = b1[j];
                    for (int f = 0; f < F; ++f) a += W1[(size_t)j * F + f] * x[f];
                    h1[j] = a > 0.0 ? a : 0.0;
                    d1[j] = 0.0;
                }
                *db3 += gr;
                for (int k = 0; k < H2; ++k) {
                    double a = b2[k];
                    for (int j = 0; j < H1; ++j) a += W2[(size_t)k * H1 + j] * h1[j];
                    h2[k] = a > 0.0 ? a : 0.0;
                    dW3[k] += gr * h2[k];
                    d2[k] = (a > 0.0) ? gr * W3[k] : 0.0;
                    db2[k] += d2[k];
                    for (int j = 0; j < H1; ++j) {
                        dW2[(size_t)k * H1 + j] += d2[k] * h1[j];
                        d1[j] += d2[k] * W2[(size_t)k * H1 + j];
                    }
                }
                for (int j = 0; j < H1; ++j) {
                    if (!(h1[j] > 0.0)) continue;
                    db1[j] += d1[j];
                    for (int f = 0; f < F; ++f) dW1[(size_t)j * F + f] += d1[j] * x[f];
                }
            }
        if (scores_out) memcpy(scores_out, scores, sizeof(double) * R);
    }
    free(scores); free(ds); free(h1); free(h2); free(d1); free(d2);
    return rc;
}

/* Listwise softmax cross-entropy (ListNet top-one).  NOT in the reference
 * (pytorchltr/loss/__init__.py:1-7 exports seven pairwise classes only): PARITY UNPINNED -- this is
 * the builder's own specification (include/ltr_hip.h: ltr_listwise_softmax_f32), restated here in
 * fp64 and checked by finite differences in tests/test_listwise.py.
 *   loss[b] = -sum_{j<n} softmax(y[b,:n])_j * ln softmax(s[b,:n])_j;  d/ds_j = P_s(j) - P_y(j). */
int oracle_listwise_softmax(const double *scores, const double *rel, const int64_t *n, int B, int L,
                            double *loss, double *dscores)
{
    for (int b = 0; b < B; ++b) {
        const double *s = scores + (size_t)b * L, *y = rel + (size_t)b * L;
        int nb = clamp_n(n[b], L);
        double ms = -INFINITY, my = -INFINITY, zs = 0.0, zy = 0.0, acc = 0.0;
        for (int j = 0; j < nb; ++j) { if (s[j] > ms) ms = s[j]; if (y[j] > my) my = y[j]; }
        for (int j = 0; j < nb; ++j) { zs += exp(s[j] - ms); zy += exp(y[j] - my); }
        for (int j = 0; j < nb; ++j)
            acc -= exp(y[j] - my) / zy * ((s[j] - ms) - log(zs));
        loss[b] = nb > 0 ? acc : 0.0;
        if (dscores)
            for (int j = 0; j < L; ++j)
                dscores[(size_t)b * L + j] = (j < nb) ? exp(s[j] - ms) / zs - exp(y[j] - my) / zy : 0.0;
    }
    return 0;
}
