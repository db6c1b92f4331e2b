# usage: bash tools/abl_build.sh <file.hip> <tag> <extra hipcc flags...>  -> rdst_amd/lib_<tag>.so: the release library with that
# one source recompiled with the extra flags (compile-time ablation variants for A/B runs on one box)
F=$1; T=$2; shift; shift
B=$(basename $F .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result "$@" -c $F -o /tmp/_abl_$T.o || exit 1
OBJS=$(ls rdst_amd/csrc/_obj/*.o | grep -v "/$B.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o rdst_amd/lib_$T.so $OBJS /tmp/_abl_$T.o
