/* The scoped device binding of every launching entry point (DeviceGuard, csrc/fpv_hip.hip) on a host with TWO devices - which the
 * builder's boxes never had: a handle created on device 1 is used while the thread's current device is 0.  The preloaded stand-in
 * runtime (fake_hip.c) records hipSetDevice; the launch itself then fails in the real runtime (there is no GPU here), which is the
 * point: even on the error path the caller's device is put back.  Prints one line of JSON for tests/test_device_guard.py. */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/fpv_abi.h"

int main(int argc, char** argv)
{
    int (*current)(void) = (int (*)(void))dlsym(RTLD_DEFAULT, "fake_hip_current_device");
    int (*calls)(int*, int) = (int (*)(int*, int))dlsym(RTLD_DEFAULT, "fake_hip_set_device_calls");
    void (*make_current)(int) = (void (*)(int))dlsym(RTLD_DEFAULT, "fake_hip_make_current");
    if (!current || !calls || !make_current) { fprintf(stderr, "run me with LD_PRELOAD=libfake_hip.so\n"); return 2; }
    fpv_params_t p;
    FILE* f = fopen(argc > 1 ? argv[1] : "", "rb");
    if (!f || fread(&p, sizeof p, 1, f) != 1) { fprintf(stderr, "cannot read the params blob\n"); return 2; }
    fclose(f);
    fpv_handle_t h1 = 0, h0 = 0, none = 0;
    if (fpv_create(&p, 4096, 1, &h1) != FPV_OK || fpv_create(&p, 4096, 0, &h0) != FPV_OK) { fprintf(stderr, "fpv_create: %s\n", fpv_last_error()); return 3; }
    if (fpv_create(&p, 4096, 2, &none) != FPV_ENODEV || none) return 4;                       /* two devices: index 2 is out of range */
    fpv_buffers_t b;
    memset(&b, 0, sizeof b);
    b.state = (float*)(uintptr_t)0x7f0000000000ull;                                 /* never dereferenced on the host; no kernel will run */
    b.action = (const float*)(uintptr_t)0x7f0010000000ull;
    b.ld = 4096 + 256;
    int log[64];
    make_current(0);
    const int before = calls(log, 64);
    const int rc1 = fpv_step(h1, &b, 0);                                            /* handle on device 1, caller on device 0 */
    const int n1 = calls(log, 64), cur1 = current();
    const int s0 = n1 > before ? log[before] : -1, s1 = n1 > before + 1 ? log[before + 1] : -1;
    const int rc0 = fpv_step(h0, &b, 0);                                            /* handle on the caller's device: no switch at all */
    const int n0 = calls(log, 64), cur0 = current();
    make_current(1);
    const int rc2 = fpv_reset(h0, &b, 0, 0, 0, 0, 0);                               /* the other way round */
    const int n2 = calls(log, 64), cur2 = current();
    printf("{\"rc_step_other_device\": %d, \"set_device_calls\": %d, \"first\": %d, \"second\": %d, \"current_after\": %d, "
           "\"rc_step_same_device\": %d, \"set_device_calls_same_device\": %d, \"current_after_same\": %d, "
           "\"rc_reset_from_device_1\": %d, \"set_device_calls_reset\": %d, \"reset_first\": %d, \"reset_second\": %d, \"current_after_reset\": %d}\n",
           rc1, n1 - before, s0, s1, cur1, rc0, n0 - n1, cur0, rc2, n2 - n0, n2 > n0 ? log[n0] : -1, n2 > n0 + 1 ? log[n0 + 1] : -1, cur2);
    fpv_destroy(h1);
    fpv_destroy(h0);
    return 0;
}
