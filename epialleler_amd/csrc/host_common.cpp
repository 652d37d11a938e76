// Pure host code shared by the HIP library and the host-only sanitizer build (`make asan`): the thread-local error
// message behind epi_last_error() and the cached environment switches (common.hpp: Options).  No HIP call in here.
#include "common.hpp"
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <mutex>

namespace epi {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

// ---- options ------------------------------------------------------------------
static Options g_options;
static std::once_flag g_options_once;

static void read_options() {
  Options o;
  auto geti = [](const char *name, int *dst) { if (const char *e = getenv(name)) *dst = atoi(e); };
  geti("EPIHIP_DEVICE", &o.device);
  geti("EPIHIP_CX_SLOT", &o.cx_slot);
  if (const char *e = getenv("EPIHIP_CX_LEAN")) o.cx_lean = atoi(e) != 0;
  geti("EPIHIP_CX_WALK", &o.cx_walk);
  geti("EPIHIP_HEAVY_ROWS", &o.heavy_rows);
  if (o.heavy_rows < 0) o.heavy_rows = 0;
  if (const char *e = getenv("EPIHIP_TILE_HINT")) o.tile_hint = atoi(e) != 0;
  geti("EPIHIP_REALIGN", &o.realign);
  if (o.realign != 0 && o.realign != 4 && o.realign != 8) o.realign = 16;
  if (const char *e = getenv("EPIHIP_MHL_FUSED")) o.mhl_fused = atoi(e) != 0;
  geti("EPIHIP_MHL_SLOT", &o.mhl_slot);
  geti("EPIHIP_MHL_WG", &o.mhl_wg);
  if (o.mhl_wg != 256 && o.mhl_wg != 512) o.mhl_wg = 0;
  geti("EPIHIP_MHL_TILE_GROUP", &o.mhl_tile_group);
  o.mhl_multi = getenv("EPIHIP_MHL_MULTI") != nullptr;
  if (const char *e = getenv("EPIHIP_MHL_GROUP")) { int g = 0, c = 0; if (sscanf(e, "%d,%d", &g, &c) == 2) { o.mhl_group_g = g; o.mhl_group_c = c; } else o.mhl_group_g = -1; }
  geti("EPIHIP_MHL_SUMS", &o.mhl_sums);
  if (const char *e = getenv("EPIHIP_MHLF_SHAPE")) {
    int g = 0, ca = 0, cb = 0;
    const int nf = sscanf(e, "%d,%d,%d", &g, &ca, &cb);
    const bool two = nf == 3 && cb == 2 && ca == 3 && g >= 4 && g <= 32;
    if (nf >= 2 && (g == 2 || g == 4 || g == 8 || g == 16 || g == 32 || g == 64) && ca >= 2 && ca <= 4 && (nf == 2 || cb == 0 || two))
      o.mhlf_shape = g * 100 + ca * 10 + (nf == 3 ? cb : 0);
  }
  if (const char *e = getenv("EPIHIP_MHLF_FOLD")) o.mhlf_fold = atoi(e) != 0;
  geti("EPIHIP_MHLF_FOLD_SLOTS", &o.mhlf_fold_slots);
  geti("EPIHIP_GROUP", &o.pr_group);
  geti("EPIHIP_PR_RPG", &o.pr_rpg);
  if (const char *e = getenv("EPIHIP_PR_WIDE")) o.pr_wide = atoi(e) != 0;
  o.bam_timing = getenv("EPIHIP_BAM_TIMING") != nullptr;
  o.no_libdeflate = getenv("EPIHIP_NO_LIBDEFLATE") != nullptr;
  o.no_hugepage = getenv("EPIHIP_NO_HUGEPAGE") != nullptr;
  g_options = o;
}

const Options &options() {
  std::call_once(g_options_once, read_options);
  return g_options;
}

}  // namespace epi

extern "C" {

const char *epi_last_error(void) { return epi::g_err; }

// test hook: the environment switches (common.hpp: Options) are read once per process; this re-reads them
void epi_options_reload(void) { (void)epi::options(); epi::read_options(); }

}  // extern "C"
