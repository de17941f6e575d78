for f in 0 8 12 0 8; do echo -n "flags $f: "; T2V_P16_FLAGS=$f timeout 100 python tools/dbg/persist16_prof.py 16 84 400 2>&1 | grep "kernel k_dec"; done
