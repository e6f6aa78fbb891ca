#!/bin/bash
# Development: group 1 (large models) rebuilt alone with extra compiler flags and linked with the objects of the last full
# build into build/lib_dev.so (select it with AGX_LIB=build/lib_dev.so).   scripts/build_dev.sh -DAGX_BLK_STAMP
set -e
cd "$(dirname "$0")/.."
G=${AGX_DEV_GROUP:-1}
FLAGS=$(python3 - <<PY
import importlib.util, pathlib
spec = importlib.util.spec_from_file_location("agx_front", "agimus_controller_amd/csrc/agx_front.py")
f = importlib.util.module_from_spec(spec); spec.loader.exec_module(f)
names = [n for _, n, _ in f.prototypes(pathlib.Path("include/agimus_hip.h").read_text())]
print(" ".join(f.rename_flags(names, $G)))
PY
)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -Iinclude $FLAGS "$@" -o build/obj/dev_g$G.o agimus_controller_amd/csrc/agimus_hip.hip
OBJS=""
for g in 0 1; do if [ $g = $G ]; then OBJS="$OBJS build/obj/dev_g$g.o"; else OBJS="$OBJS build/obj/agx_g$g.o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/lib_dev.so $OBJS build/obj/agx_front.o
echo built build/lib_dev.so
