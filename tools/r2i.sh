#!/bin/bash
python bench.py --no-cpu-baseline | cut -c60-140
MSSEG_K3WG_GXCAP=8 MSSEG_K3WG_GXCAP_TILES=200 python bench.py --no-cpu-baseline | cut -c60-140
MSSEG_K3WG_GXCAP=16 MSSEG_K3WG_GXCAP_TILES=200 python bench.py --no-cpu-baseline | cut -c60-140
MSSEG_K3WG_GXCAP=8 MSSEG_K3WG_GXCAP_TILES=20 python bench.py --no-cpu-baseline | cut -c60-140
python bench.py --no-cpu-baseline | cut -c60-140
