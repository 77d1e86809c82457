// knobs.hpp — which environment variables the library reads (round 5, VERDICT r4 item 5 / ADVICE r4).
//
// PRODUCT build (build.sh): the library reads only the OPERATIONAL variables below, each of them once - at context creation (into
// svgp_ctx::kn) or on first use for the process-wide ones - and never on the per-evaluation path.  Every tuning knob, A/B switch and
// rejected variant of rounds 1-5 is a compile-time constant (its default) and the code only such a knob could reach is not compiled.
//
//   SVGP_TIMING, SVGP_OVERLAP, SVGP_SEG_SPLIT      per context (svgp_ctx_create)
//   SVGP_DEBUG_SYNC, SVGP_RCCL_LIB, SVGP_DISABLE_RCCL      per process (first use)
//   SVGP_OFFLOAD_MIN_WORK      by svgp_offload_advice, a host-side advisory function outside every evaluation (hosts change it at run time)
//
// EXPERIMENTS build (tools/build_experiments.sh: -DSVGP_EXPERIMENTS, libsvgp_experiments.so): the same sources with every knob
// read from the environment (per context where a test toggles it, else once per process) and the rejected variants compiled in; it
// exports the extra symbol svgp_debug_experiments so that tools and tests can tell the two apart.  INTEGRATION.md lists both sets.
#pragma once
#include <cstdlib>

namespace svgp {

#ifdef SVGP_EXPERIMENTS
constexpr bool kExperiments = true;
inline int exp_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
inline long long exp_ll(const char* name, long long dflt) { const char* e = getenv(name); return e ? atoll(e) : dflt; }
inline double exp_double(const char* name, double dflt) { const char* e = getenv(name); return e ? atof(e) : dflt; }
#else
constexpr bool kExperiments = false;
constexpr int exp_int(const char*, int dflt) { return dflt; }
constexpr long long exp_ll(const char*, long long dflt) { return dflt; }
constexpr double exp_double(const char*, double dflt) { return dflt; }
#endif

// operational variables: read by every build.  Strictly "0" or "1": anything else (empty, "on", "true", "2") keeps the default - atoi
// turned SVGP_TIMING=on into 0 and silently switched the timing off
inline int env_flag(const char* name, int dflt) {
  const char* e = getenv(name);
  if (!e || (e[0] != '0' && e[0] != '1') || e[1] != '\0') return dflt;
  return e[0] - '0';
}

// Per-context settings, filled by read_knobs() in svgp_ctx_create.  The first three are operational; the rest keep their defaults
// in the product build.
struct Knobs {
  int timing = 1;            // SVGP_TIMING: 0 = no timing events on the stream (svgp_last_timing reports zeros)
  int overlap = 1;           // SVGP_OVERLAP: 0 = never run strips beside the factorisation
  int seg_split = 1;         // SVGP_SEG_SPLIT: 0 = never split the closing launch of a small overlapped batch (bitwise the serial kernel)
  // ---- experiments build only ----
  int overlap_min_panels = 5;        // SVGP_OVERLAP_MIN_PANELS
  int overlap_head = 0;              // SVGP_OVERLAP_HEAD: segmented head of multi-round batches (measured: no gain)
  int overlap_head_min_panels = 4;   // SVGP_OVERLAP_HEAD_MIN_PANELS
  int overlap_dry = 0;               // SVGP_OVERLAP_DRY: enqueue the segments without the prep (timing experiments)
  int syrk_uniform = 1;              // SVGP_SYRK_UNIFORM: 0 = the weighted SYRK for the Gaussian likelihood too
  int pipe_lanes = 1, pipe_streams = 1, pipe_prio = 0, pipe_mode = 1;   // SVGP_GRAD_PIPELINE / _PIPE_STREAMS / _PIPE_PRIO / _PIPE_MODE (round 5: measured, rejected)
};

inline void read_knobs(Knobs& k) {
  k.timing = env_flag("SVGP_TIMING", 1) != 0;
  k.overlap = env_flag("SVGP_OVERLAP", 1) != 0;
  k.seg_split = env_flag("SVGP_SEG_SPLIT", 1) != 0;
#ifdef SVGP_EXPERIMENTS
  k.overlap_min_panels = exp_int("SVGP_OVERLAP_MIN_PANELS", 5);
  k.overlap_head = exp_int("SVGP_OVERLAP_HEAD", 0) == 1;
  k.overlap_head_min_panels = exp_int("SVGP_OVERLAP_HEAD_MIN_PANELS", 4);
  k.overlap_dry = exp_int("SVGP_OVERLAP_DRY", 0);
  k.syrk_uniform = exp_int("SVGP_SYRK_UNIFORM", 1) != 0;
  k.pipe_lanes = exp_int("SVGP_GRAD_PIPELINE", 1);
  k.pipe_lanes = k.pipe_lanes < 2 ? 1 : (k.pipe_lanes > 4 ? 4 : k.pipe_lanes);
  k.pipe_streams = exp_int("SVGP_GRAD_PIPE_STREAMS", 1);
  k.pipe_streams = k.pipe_streams < 1 ? 1 : (k.pipe_streams > 2 ? 2 : k.pipe_streams);
  if (k.pipe_streams > k.pipe_lanes) k.pipe_streams = k.pipe_lanes;
  k.pipe_prio = exp_int("SVGP_GRAD_PIPE_PRIO", 0);
  k.pipe_mode = exp_int("SVGP_GRAD_PIPE_MODE", 1);
#endif
}

}  // namespace svgp
