#!/bin/bash
# dev: register / LDS / spill figures of the sweep kernels (device-only compile)
# usage: tools/kres.sh [fast|exact] [extra -D flags]
mode=${1:-fast}; shift
if [ "$mode" = exact ]; then F="-DVM_EXACT=1 -ffp-contract=off"; else F="-DVM_EXACT=0 -ffp-contract=off"; fi   # the flags of videomorphing_amd/build.py
src=${SRC:-videomorphing_amd/csrc/vm_sweep_kernels.hip}
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 --offload-device-only -Iinclude $F "$@" -c $src -o /tmp/kres_$$.co || exit 1
python3 -c "import sys; d=open(sys.argv[1],'rb').read(); open(sys.argv[1],'wb').write(d[d.find(b'\x7fELF'):])" /tmp/kres_$$.co
/opt/rocm/lib/llvm/bin/llvm-readelf --notes /tmp/kres_$$.co | awk '/\.name:/{n=$2} /\.vgpr_count:/{v=$2} /\.vgpr_spill_count:/{s=$2} /\.group_segment_fixed_size:/{l=$2} /\.private_segment_fixed_size:/{p=$2} /\.sgpr_count:/{g=$2} /\.wavefront_size:/{print n, "vgpr="v, "sgpr="g, "spill="s, "scratch="p, "lds="l}'
rm -f /tmp/kres_$$.co
