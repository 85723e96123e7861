// Host build of the PRODUCT's elementary functions (include/fh_elementary.h) for tests/test_oracle_anchors.py: the CPU checker has its own implementation of
// the same numerical specification (oracle/oelementary.h) and the two must return the same bits.  Test scaffolding; kinds as oracle/pyoracle.py ELEMENTARY.
#include "../../include/fh_elementary.h"

extern "C" void fhe_host(int kind, int n, const float* x, const float* y, float* out)
{
  for (int i = 0; i < n; ++i) {
    switch (kind) {
      case 0: out[i] = fhe_sin(x[i]); break;
      case 1: out[i] = fhe_cos(x[i]); break;
      case 2: out[i] = fhe_exp(x[i]); break;
      case 3: out[i] = fhe_log(x[i]); break;
      case 4: out[i] = fhe_pow(x[i], y[i]); break;
      case 5: out[i] = fhe_acos(x[i]); break;
      case 6: out[i] = fhe_atan2(x[i], y[i]); break;
      case 7: out[i] = fhe_log2(x[i]); break;
      case 8: out[i] = fhe_pow1p5(x[i]); break;
      default: out[i] = 0.0f;
    }
  }
}
