#!/bin/bash
# dev: rebuild the library ON THE GPU BOX with extra defines and run a command with it; the in-tree build is restored after
# usage: bash tools/gpu_variant.sh "<defines>" <command...>
defs=$1; shift
cp videomorphing_amd/lib/libvmorph_hip.so /tmp/libvmorph_keep.so
VM_DEFS="$defs" python3 -c "
import os
from videomorphing_amd import build
print('COMMON', build.COMMON[:3])
print(build.build(force=True))" > /tmp/variant_build.log 2>&1 || { tail -5 /tmp/variant_build.log; exit 1; }
tail -2 /tmp/variant_build.log >&2; md5sum videomorphing_amd/lib/libvmorph_hip.so >&2
"$@"
cp /tmp/libvmorph_keep.so videomorphing_amd/lib/libvmorph_hip.so
