for rep in 1 2; do
python tools/kuf_time.py H H32 C3 C4 | grep kuf
SVGP_KUF_PERSIST=0 python tools/kuf_time.py H H32 C3 C4 | grep kuf | sed 's/^/  np /'
SVGP_KUF_V1=1 python tools/kuf_time.py H H32 C3 C4 | grep kuf | sed 's/^/  v1 /'
done
