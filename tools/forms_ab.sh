# alternating processes over library builds: tools/forms_ab.sh <outfile> <rounds> <tag> [<tag> ...]   (tag "product" = the in-tree library,
# others = tools/libopenimpala_amd_<tag>.so from tools/variant_build.sh)
out=$1; rounds=$2; shift 2
mkdir -p "$(dirname "$out")"; : > "$out"
for r in $(seq 1 $rounds); do
  for tag in "$@"; do
    if [ "$tag" = product ]; then lib=$PWD/openimpala_amd/libopenimpala_amd.so; else lib=$PWD/tools/libopenimpala_amd_$tag.so; fi
    OI_LIB_PATH=$lib timeout -k 10 300 python tools/forms_ab.py ${FORMS_N:-1024} 2 ${FORMS_CG1:-1} >> "$out" 2>&1 || { echo "FAILED $tag" >> "$out"; tail -5 "$out"; exit 1; }
  done
done
sort -s -k1,1 "$out" | grep -v digest; grep digest "$out" | sort | uniq -c
