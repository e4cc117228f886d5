#!/bin/bash
# random-access ceilings of the chip for the search kernel's access shapes (tools/gather_probe.hip)
P=tools/gather_probe
for mib in 128 1024 8192; do for rb in 128 512; do $P $mib $rb 16 8 0; done; done
$P 1024 128 16 4 0; $P 1024 128 8 8 0; $P 1024 64 16 8 0
for mode in 1 2 3; do for mib in 32 512 4096; do $P $mib 128 16 4 $mode; done; done
