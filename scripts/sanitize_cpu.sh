#!/bin/bash
# AddressSanitizer + UBSan over the CPU side: the C oracle, the device header compiled for the host
# (tests/emu) and the host flavour of the ABI (csrc/gobblet_cpu.cpp, round 5), driven by the CPU test-suite.  GPU sanitizers are not available on the pool; the device
# code's indexing is exercised here through the same header.   usage: scripts/sanitize_cpu.sh
set -e
cd "$(dirname "$0")/.."
T=$(mktemp -d)
gcc -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -shared -fPIC -pthread -o $T/oracle.so oracle/gobblet_oracle.c
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -Wno-unknown-pragmas -Wno-attributes -shared -fPIC \
    -o $T/emu.so tests/emu/emu_device.cpp
g++ -O1 -g -std=c++17 -mpopcnt -fsanitize=address,undefined -fno-omit-frame-pointer -Wno-unknown-pragmas -Wno-attributes -shared -fPIC -pthread \
    -o $T/cpu.so gobblet-rl_amd/csrc/gobblet_cpu.cpp
python -c "import oracle; oracle.lib(); from tests import emu; emu.lib(); import gobblet_rl_amd as G; G._native.build_cpu()"   # make sure the normal builds exist
C=gobblet-rl_amd/csrc/libgobblet_cpu.so
cp oracle/libgobblet_oracle.so $T/oracle.bak; cp tests/emu/libgobblet_emu.so $T/emu.bak; cp $C $T/cpu.bak
restore() { cp $T/oracle.bak oracle/libgobblet_oracle.so; cp $T/emu.bak tests/emu/libgobblet_emu.so; cp $T/cpu.bak $C; touch oracle/libgobblet_oracle.so tests/emu/libgobblet_emu.so $C; }
trap restore EXIT
cp $T/oracle.so oracle/libgobblet_oracle.so; cp $T/emu.so tests/emu/libgobblet_emu.so; cp $T/cpu.so $C
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0 \
    python -m pytest tests/test_oracle_golden.py tests/test_device_emulation.py tests/test_properties.py tests/test_sharding_gloo.py tests/test_cpu_twin.py \
    -q -m "not gpu" > $T/log.txt 2>&1 || true
tail -2 $T/log.txt
echo "sanitizer reports: $(grep -c 'runtime error\|AddressSanitizer' $T/log.txt || true)"
