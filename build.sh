#!/usr/bin/env bash
# Same role as the reference's build.sh (hw8/build.sh: cmake Release): builds librtamd.so + the CLI for gfx950.
set -e
cd "$(dirname "$0")"
make -C raytracing-course-hw_amd/csrc -j"$(nproc)"
mkdir -p build
ln -sf ../raytracing-course-hw_amd/rtamd_main build/main
