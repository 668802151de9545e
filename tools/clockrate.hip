#include <hip/hip_runtime.h>
#include <cstdio>
int main(){int r=0; hipDeviceGetAttribute(&r, hipDeviceAttributeWallClockRate, 0); printf("wall clock rate kHz: %d\n", r);
int c=0; hipDeviceGetAttribute(&c, hipDeviceAttributeClockRate, 0); printf("clock rate kHz: %d\n", c); return 0;}
