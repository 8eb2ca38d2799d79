// Library-level entry points of libpasnl_hip.so (see include/pasnl.h).
#include "common.hpp"

extern "C" int pasnl_version(void) { return PASNL_VERSION; }

extern "C" const char* pasnl_strerror(int code) {
  switch (code) {
    case PASNL_OK: return "ok";
    case PASNL_EINVAL: return "invalid argument (shape or attribute)";
    case PASNL_ENULL: return "null pointer";
    case PASNL_EWORKSPACE: return "workspace too small";
    case PASNL_ELAUNCH: return "HIP launch failed";
    case PASNL_EUNSUPPORTED: return "request outside the supported range of the gfx950 kernels";
    default: return "unknown pasnl error";
  }
}

extern "C" int pasnl_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return -1;
  }
  return n;
}
