#!/bin/bash
# builds wakeword_trainer_home_amd/csrc/libwwhip_ab.so = the current objects, except the named source taken from a git revision
# (default HEAD) or compiled from the working tree with extra flags:   tools/build_ab.sh ww_frontend [HEAD|-DWW_MACRO ...]
set -e
cd "$(dirname "$0")/../wakeword_trainer_home_amd/csrc"
src=$1; shift
what=${1:-HEAD}
tmp=$(mktemp -d)
flags="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -I../../include -I."
if [[ $what == -* ]]; then
  hipcc $flags "$@" -c $src.hip -o $tmp/$src.o
else
  git show "$what:wakeword_trainer_home_amd/csrc/$src.hip" > $tmp/$src.hip
  hipcc $flags -c $tmp/$src.hip -o $tmp/$src.o
fi
objs=""
for o in *.o; do [[ $o == $src.o ]] && objs="$objs $tmp/$src.o" || objs="$objs $o"; done
hipcc -shared -fPIC --offload-arch=gfx950 -o libwwhip_ab.so $objs
rm -rf $tmp
echo "libwwhip_ab.so: $src from $what $*"
