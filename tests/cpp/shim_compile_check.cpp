// Compile-only check (g++ -fsyntax-only): the drop-in headers agree with the C ABI layouts.
#include "SimplePathtracer.h"
static_assert(sizeof(Material) == 104, "");
static_assert(sizeof(LaunchParams) == 248, "");
static_assert(sizeof(Probe) == 64, "");
static_assert(offsetof(LaunchParams, frame.c) == 72, "");
static_assert(offsetof(LaunchParams, frame.offset) == 88, "");
static_assert(offsetof(LaunchParams, samples_per_launch) == 152, "");
int main() { return 0; }
