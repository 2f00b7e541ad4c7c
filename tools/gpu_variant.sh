#!/bin/bash
# dev: build the library ON THE GPU BOX with extra defines and run a command with THAT library.  The variant gets
# object and library directories of its own (videomorphing_amd/build_<hash>, lib_<hash>: build.py), so the in-tree
# product build is never touched and nothing has to be restored.
# usage: bash tools/gpu_variant.sh "<defines>" <command...>
defs=$1; shift
lib=$(VM_DEFS="$defs" python3 -c "
from videomorphing_amd import build
print(build.build())" 2> /tmp/variant_build.log | tail -1) || { tail -5 /tmp/variant_build.log; exit 1; }
[ -f "$lib" ] || { echo "variant build failed"; tail -5 /tmp/variant_build.log; exit 1; }
echo "variant library: $lib" >&2
VM_LIB_PATH="$lib" "$@"
