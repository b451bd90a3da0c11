/* A stand-in for the four HIP runtime calls that fpv_create / fpv_device_cache_model make, preloaded (LD_PRELOAD) in front of
 * libamdhip64 by tests/test_device_guard.py: the device's answers come from the environment, so the library's reaction to a device
 * that is NOT the one its cache model was measured on - a CPX compute partition, another architecture - is tested without owning
 * such a device (and without any GPU).  No kernel is launched through it: the driver only creates handles and asks them. */
#include <stdlib.h>
#include <string.h>

#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

static int env_int(const char* k, int dflt) { const char* v = getenv(k); return v && *v ? atoi(v) : dflt; }

/* the thread's current device and every hipSetDevice call, so that a driver can watch the library's scoped device binding
 * (DeviceGuard in csrc/fpv_hip.hip) switch to a handle's device and put the caller's back */
static int g_current = 0, g_calls[64], g_ncalls = 0;
int fake_hip_current_device(void) { return g_current; }
int fake_hip_set_device_calls(int* out, int max) { for (int i = 0; i < g_ncalls && i < max; ++i) out[i] = g_calls[i]; return g_ncalls; }
void fake_hip_make_current(int device) { g_current = device; }

hipError_t hipGetDeviceCount(int* count) { *count = env_int("FAKE_HIP_DEVICES", 1); return hipSuccess; }
hipError_t hipGetDevice(int* device) { *device = g_current; return hipSuccess; }
hipError_t hipSetDevice(int device)
{
    if (device < 0 || device >= env_int("FAKE_HIP_DEVICES", 1)) return hipErrorInvalidDevice;
    if (g_ncalls < 64) g_calls[g_ncalls++] = device;
    g_current = device;
    return hipSuccess;
}

hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_tR0600* prop, int device)
{
    (void)device;
    memset(prop, 0, sizeof(*prop));
    const char* arch = getenv("FAKE_HIP_ARCH");
    strncpy(prop->gcnArchName, arch && *arch ? arch : "gfx950:sramecc+:xnack-", sizeof(prop->gcnArchName) - 1);
    prop->multiProcessorCount = env_int("FAKE_HIP_CUS", 256);
    prop->l2CacheSize = env_int("FAKE_HIP_L2", 4 << 20);
    return hipSuccess;
}
