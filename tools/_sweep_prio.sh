run() { echo "== [$1] $2"; TC_OPTS="$1" timeout 100 python tools/_time_cases.py $2 2>&1 | grep -v amdgpu.ids; }
for o in "" "108=4,113=20" "108=4,113=4" "108=3,113=4" "108=2,113=20" "108=3,113=20" ""; do run "$o" "1280x720:1 640x480:3 640x480:4 640x480:5 640x480:6"; done
