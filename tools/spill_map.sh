# usage: bash tools/spill_map.sh <file.hip> <kernel-name-regex>   -> where in the ISA of that kernel the scratch accesses are
F=$1; K=$2
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o /tmp/_sm.s $F -I$(dirname $F) 2>/dev/null
a=$(grep -n "^_ZN.*$K.*:" /tmp/_sm.s | head -1 | cut -d: -f1)
sed -n "$a,\$p" /tmp/_sm.s | awk '/s_endpgm/{print; exit} {print}' > /tmp/_k.s
wc -l /tmp/_k.s
awk '
function flush(tag){ if (sl+ss>1 || mf>0) printf "%d %s sl=%d ss=%d mfma=%d exp=%d ds=%d gl=%d gs=%d valu=%d\n", NR, tag, sl, ss, mf, ex, ds, gl, gs, va; sl=ss=mf=ex=ds=gl=gs=va=0}
/s_barrier/ {flush("BARRIER")}
/^\.LBB[0-9_]+:/ {flush($1)}
/scratch_load/ {sl++} /scratch_store/ {ss++} /v_mfma/ {mf++} /v_exp_f32/ {ex++} /ds_read|ds_load/ {ds++} /global_load/ {gl++} /global_store/ {gs++} /^\tv_/ {va++}
' /tmp/_k.s
