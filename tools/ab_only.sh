for r in 1 2 3 4 5 6; do for d in . _old; do (cd $d; python bench_sequence.py --frames 120 --quiet 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); s=j['steady_state']; print('$d', round(s['scans_per_s'],1), round(s['median_process_frame_ms'],4), round(s['median_mapping_ms'],4))
"); done; done
