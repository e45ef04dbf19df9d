// Library runtime: version string and thread-local error plumbing of the C ABI.
#include <string.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include "common.h"

namespace hoisdf {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return HOISDF_ERR_LAUNCH;
  }
  return HOISDF_OK;
}

static int g_det = -1;          // -1: not decided yet (environment), 0 / 1
static int g_gemm_emu = 1;      // ... as fp32 emulated on the bf16 pipe (gemm_emu.hip); default on, HOISDF_GEMM=f32 / the setter turn it off

bool gemm_emu_mode() {
  static int env = -1;
  if (env < 0) {
    const char* e = getenv("HOISDF_GEMM");
    env = (e && strcmp(e, "f32") == 0) ? 0 : 1;          // exactly "f32" -> off (hoisdf_amd/ops.py parses the same string)
    if (!env) g_gemm_emu = 0;
  }
  return g_gemm_emu != 0;
}

bool deterministic_mode() {
  if (g_det < 0) {
    const char* e = getenv("HOISDF_DETERMINISTIC");
    g_det = (e && e[0] && e[0] != '0') ? 1 : 0;
  }
  return g_det == 1;
}

DetScratch det_scratch(size_t floats) {
  static float* part = nullptr;
  static unsigned* ticket = nullptr;
  static size_t cap = 0;
  DetScratch d{nullptr, nullptr, 0};
  if (!deterministic_mode()) return d;
  if (floats > cap || !ticket) {
    // grown on demand, never freed (a few MB); deterministic mode runs on one stream, so launches never overlap
    if (part) (void)hipFree(part);
    cap = floats < (1u << 20) ? (1u << 20) : floats;
    if (hipMalloc(&part, cap * sizeof(float)) != hipSuccess) { part = nullptr; cap = 0; return d; }
    if (!ticket) {
      if (hipMalloc(&ticket, sizeof(unsigned)) != hipSuccess) { ticket = nullptr; return d; }
      (void)hipMemset(ticket, 0, sizeof(unsigned));
    }
  }
  d.part = part; d.ticket = ticket; d.on = 1;
  return d;
}

}  // namespace hoisdf

extern "C" void hoisdf_set_gemm_emu(int on) { (void)hoisdf::gemm_emu_mode(); hoisdf::g_gemm_emu = on ? 1 : 0; }
extern "C" int hoisdf_get_gemm_emu(void) { return hoisdf::gemm_emu_mode() ? 1 : 0; }
extern "C" void hoisdf_set_deterministic(int on) { hoisdf::g_det = on ? 1 : 0; }
extern "C" int hoisdf_get_deterministic(void) { return hoisdf::deterministic_mode() ? 1 : 0; }
extern "C" const char* hoisdf_version(void) { return "hoisdf-hip 0.1 (gfx950)"; }
extern "C" const char* hoisdf_last_error(void) { return hoisdf::g_err; }
