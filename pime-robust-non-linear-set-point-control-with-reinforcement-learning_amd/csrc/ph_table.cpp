// Titration curve of the pH-neutralisation process -- host code (sequential warm-started Newton, fp64).
// replaces: PH1D.__init__, /root/reference/gym_control/envs/ph.py:72-84 (500 000 interpreted Newton steps).
#include <cmath>

#include "pime_common.hpp"

extern "C" int pime_ph_table_build(const pime_ph_chem* chem, double mhcl_step, int32_t n, double* out) {
    PIME_REQUIRE(out != nullptr && n > 0 && mhcl_step > 0, "pime_ph_table_build: bad arguments (n=%d)", n);
    const pime_ph_chem dflt{1e-14, 5.6e-10, 0.5e-5, 0.01, 0.005, 0.01};  // ph.py:32-37
    const pime_ph_chem c = chem ? *chem : dflt;
    const double kk = c.kchem + c.ka;
    double H = 1e-14 / c.MNaOH;  // ph.py:75; every later entry starts from its predecessor's root
    for (int32_t i = 0; i < n; ++i) {
        const double m = (double)i * mhcl_step;
        // [H+] solves H^4 + ak H^3 + bk H^2 + ck H + dk = 0 (charge balance; ph.py:77-80)
        const double ak = c.MNH3 - m + c.MNaOH + c.kchem + c.ka;
        const double bk = kk * c.MNaOH - kk * m - c.kw + c.MNH3 * c.ka + c.kchem * c.ka - c.ka * c.MHA;
        const double ck = c.MNaOH * c.kchem * c.ka - c.kw * (c.ka + c.kchem) - m * c.kchem * c.ka - c.ka * c.kchem * c.MHA;
        const double dk = -c.kchem * c.ka * c.kw;
        for (int it = 0; it < 5; ++it) {
            // libm pow, not Horner: the reference evaluates H**4, H**3, H**2 separately and the iteration is
            // not run to convergence, so the operation order is part of the result
            const double h2 = std::pow(H, 2), h3 = std::pow(H, 3), h4 = std::pow(H, 4);
            const double f = h4 + ak * h3 + bk * h2 + ck * H + dk;
            const double df = 4 * h3 + 3 * ak * h2 + 2 * bk * H + ck;
            H = std::fabs(H - f / df);
        }
        out[i] = -1 * std::log10(H);
    }
    return PIME_OK;
}
