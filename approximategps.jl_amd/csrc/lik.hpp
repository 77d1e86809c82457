// lik.hpp — likelihood device functions shared by expect_kernel (strip.hip) and the gradient path (grad.hip):
// log p(y|f), its expectation under N(mu, v) and the derivatives of that expectation  [GPLikelihoods].
#pragma once
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace svgp {


__device__ __forceinline__ double softplus_d(double s) { return fmax(s, 0.0) + log1p(exp(-fabs(s))); }

// log Phi(x) and the hazard phi(x) / Phi(x) of the standard normal (the NormalCDFLink Bernoulli), through the scaled
// complementary error function on the negative side: no underflow and no cancellation in either tail
__device__ __forceinline__ double log_ndtr_d(double x) {
  const double u = -0.70710678118654752440 * x;   // Phi(x) = erfc(u) / 2
  return u > 0.0 ? log(0.5 * erfcx(u)) - u * u : log1p(-0.5 * erfc(-u));
}
__device__ __forceinline__ double ndtr_hazard_d(double x) {
  const double u = -0.70710678118654752440 * x;
  return u > 0.0 ? 0.79788456080286535588 / erfcx(u) : 0.39894228040143267794 * exp(-u * u) / (1.0 - 0.5 * erfc(-u));
}

// digamma: recurrence up to x >= 6, then the asymptotic series (|error| < 1e-14)
__host__ __device__ inline double digamma_d(double x) {
  double r = 0.0;
  while (x < 6.0) {
    r -= 1.0 / x;
    x += 1.0;
  }
  const double f = 1.0 / (x * x);
  return r + log(x) - 0.5 / x -
         f * (1.0 / 12.0 - f * (1.0 / 120.0 - f * (1.0 / 252.0 - f * (1.0 / 240.0 - f * (1.0 / 132.0)))));
}

// log p(y | f)  [GPLikelihoods]
__device__ __forceinline__ double loglik_point(int lik, double f, double y, double sigma2, double log_sigma2) {
  if (lik == 0) {
    const double r = y - f;
    return -0.5 * (1.8378770664093453 + log_sigma2 + r * r / sigma2);
  }
  if (lik == 1) return -softplus_d(y > 0.5 ? -f : f);
  if (lik == 2) return y * f - exp(f) - lgamma(y + 1.0);
  if (lik == 3) return -f - y * exp(-f);                                         // Exponential(scale e^f) = Gamma(1, scale e^f)
  if (lik == 5) return log_ndtr_d(y > 0.5 ? f : -f);                             // Bernoulli(Phi(f)): log Phi(+-f)
  return (sigma2 - 1.0) * log(y) - y * exp(-f) - sigma2 * f - lgamma(sigma2);    // Gamma(alpha = sigma2, scale e^f)
}

// E_{N(mu, v)}[log p(y|f)]: closed form (gh_n == 0) or Gauss-Hermite  [GPLikelihoods.expected_loglikelihood]
__device__ __forceinline__ double expected_loglik_point(const LikParams& lp, double mu, double v, double y,
                                                        double log_sigma2) {
  if (lp.gh_n == 0) {
    if (lp.lik == 0) {
      const double r = y - mu;
      return -0.5 * (1.8378770664093453 + log_sigma2 + (r * r + v) / lp.sigma2);
    }
    if (lp.lik == 2) return y * mu - exp(mu + 0.5 * v) - lgamma(y + 1.0);  // Poisson, exp link
    if (lp.lik == 3) return -mu - y * exp(0.5 * v - mu);                   // Exponential (scale), exp link
    return (lp.sigma2 - 1.0) * log(y) - y * exp(0.5 * v - mu) - lp.sigma2 * mu - lgamma(lp.sigma2);  // Gamma, exp link
  }
  const double s = 1.4142135623730951 * sqrt(v);
  double acc = 0.0;
  for (int q = 0; q < lp.gh_n; ++q) acc += lp.gh_w[q] * loglik_point(lp.lik, s * lp.gh_x[q] + mu, y, lp.sigma2, log_sigma2);
  return acc;  // weights are pre-divided by sqrt(pi)
}


// d log p(y|f) / df
__device__ __forceinline__ double dloglik_point(int lik, double f, double y, double sigma2) {
  if (lik == 0) return (y - f) / sigma2;
  if (lik == 1) return y - 1.0 / (1.0 + exp(-f));
  if (lik == 2) return y - exp(f);
  if (lik == 3) return y * exp(-f) - 1.0;
  if (lik == 5) return y > 0.5 ? ndtr_hazard_d(f) : -ndtr_hazard_d(-f);
  return y * exp(-f) - sigma2;
}

// (dE/dmu, dE/dv, dE/dsigma2) of expected_loglik_point: closed forms, or Gauss-Hermite with
// dE/dmu = sum w g'(f_q), dE/dv = sum w g'(f_q) x_q / sqrt(2 v)
__device__ __forceinline__ void expected_loglik_grad_point(const LikParams& lp, double mu, double v, double y, double& gmu,
                                                           double& gv, double& gs2) {
  gs2 = 0.0;
  if (lp.gh_n == 0) {
    if (lp.lik == 0) {
      const double r = y - mu;
      gmu = r / lp.sigma2;
      gv = -0.5 / lp.sigma2;
      gs2 = -0.5 * (1.0 / lp.sigma2 - (r * r + v) / (lp.sigma2 * lp.sigma2));
    } else if (lp.lik == 2) {
      const double e = exp(mu + 0.5 * v);
      gmu = y - e;
      gv = -0.5 * e;
    } else if (lp.lik == 3) {
      const double e = y * exp(0.5 * v - mu);
      gmu = e - 1.0;
      gv = -0.5 * e;
    } else {
      const double e = y * exp(0.5 * v - mu);
      gmu = e - lp.sigma2;
      gv = -0.5 * e;
      gs2 = log(y) - mu - lp.digamma_alpha;      // d/d alpha
    }
    return;
  }
  const double s = 1.4142135623730951 * sqrt(v);
  const double inv_s = s > 0.0 ? 1.0 / s : 0.0;
  gmu = 0.0;
  gv = 0.0;
  for (int q = 0; q < lp.gh_n; ++q) {
    const double f = s * lp.gh_x[q] + mu;
    const double dl = dloglik_point(lp.lik, f, y, lp.sigma2);
    gmu += lp.gh_w[q] * dl;
    gv += lp.gh_w[q] * dl * lp.gh_x[q] * inv_s;
    if (lp.lik == 0) gs2 += lp.gh_w[q] * (-0.5 / lp.sigma2 + 0.5 * (y - f) * (y - f) / (lp.sigma2 * lp.sigma2));
    if (lp.lik == 4) gs2 += lp.gh_w[q] * (log(y) - f - lp.digamma_alpha);
  }
}

}  // namespace svgp
