// Host orchestration + C ABI of the MI355X-native SMC++ E-step engine.
//
// Mirrors the reference's InferenceManager (include/inference_manager.h, src/inference_manager.cpp):
//   ctor / map_obs / fill_targets / populate_emission_probs  -> Engine::Engine   (21-54, 180-211, 232-254)
//   Estep                                                    -> Engine::estep    (108-114)
//   TransitionBundle::update(T, true)                        -> Engine::host_prep (src/transition_bundle.cpp:3-61)
//   loglik / Q / getters                                     -> smcpp_loglik / smcpp_q / smcpp_get_*  (116-177)
// The hot path (HMM::Estep, src/hmm.cpp:45-153) runs in the HIP kernels of kernels.hpp.  There is no CPU fallback:
// every compute entry point fails with an error if no HIP device is usable.
#include <hip/hip_runtime.h>
#include <omp.h>
#include <dlfcn.h>
#include <mutex>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/smcpp_engine.h"
#include "engine_options.hpp"   // every SMCPP_* environment switch, parsed once (the only getenv of the engine)
#include "kernels.hpp"
#include "chains2.hpp"
#include "chains_lock.hpp"
#include "chains_ss.hpp"
#include "nonsym_eig.hpp"
#include "nonsym_eig_team.hpp"
#include "prep.hpp"
#include "prep_dev.hpp"
#include "jcsfs.hpp"

// The engine is one translation unit in seven parts (each part sees everything above it):
#include "engine_base.hpp"        // logging, pinned arena, device buffers, the device route of the cold preparation (DevPrep, TwoPopDevCsfs)
#include "engine_manager.hpp"     // the manager (struct smcpp_im): observation layout, chunks, slabs, device allocation
#include "engine_params.hpp"      // parameters: cold preparation (host and device routes), device Q / gradient, uploads
#include "engine_plans.hpp"       // launch plans: chain families, the scan chains' fixed point, statistics, the E-step
#include "engine_capi.hpp"        // the C ABI of include/smcpp_engine.h
#include "engine_rccl.hpp"        // the engine-issued exchange through RCCL's C API (opt-in)
#include "engine_hostapi.hpp"     // host-only and debug exports used by the test-suite
#include "shaping.hpp"            // SURVEY.md 8 f-2 on the device: thin_data / bin_observations / compress_repeated_obs (integer, HBM-bound)
