/* CPU accuracy check of the table-based exp(-s) used by the pair kernel (same constants, same operation order,
 * hardware FMA) against a quad-precision reference.
 *   python3 tools/gen_fast_exp.py /tmp/exp_tab.h && gcc -O2 -mfma -I/tmp tools/ubench/exp_accuracy.c -lquadmath -lm -o /tmp/exp_acc && /tmp/exp_acc */
#include <stdio.h>
#include <math.h>
#include <stdlib.h>
#include <quadmath.h>
#include "exp_tab.h"
static double exp_neg(double s) {
    double n = rint(s * -EXP_INV_C);
    double r = fma(n, -EXP_C_HI, -s);
    r = fma(n, -EXP_C_LO, r);
    int ni = (int)n;
    double T = EXP_TAB[ni & (EXP_N - 1)];
    double q = fma(r, 1.0 / 6.0, 0.5);
    q = fma(r, q, 1.0);
    return ldexp(fma(T, r * q, T), ni >> EXP_BITS);
}
int main(void) {
    srand48(1);
    double maxu = 0, maxg = 0, sum = 0, sumg = 0;
    int N = 4000000;
    for (int k = 0; k < N; k++) {
        double s = drand48() * (k % 4 == 0 ? 700 : 40);
        __float128 ex = expq(-(__float128)s);
        double ref = (double)ex, u = ldexp(1.0, ilogb(ref) - 52);
        double e1 = (double)(((__float128)exp_neg(s) - ex) / u), eg = (double)(((__float128)exp(-s) - ex) / u);
        if (fabs(e1) > maxu) maxu = fabs(e1);
        if (fabs(eg) > maxg) maxg = fabs(eg);
        sum += e1; sumg += eg;
    }
    printf("table exp: max %.3f ulp, mean bias %.4f ulp | libm exp: max %.3f ulp, mean bias %.4f ulp\n", maxu, sum / N, maxg, sumg / N);
    printf("exp_neg(0)=%a exp_neg(800)=%a exp_neg(1e300)=%a\n", exp_neg(0), exp_neg(800), exp_neg(1e300));
    return 0;
}
