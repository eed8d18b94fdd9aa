"""Instruction mix of one kernel in a hipcc -S dump:  python scripts/isa_stats.py /tmp/api.s <mangled-name-substring>"""
import re, sys
s = open(sys.argv[1]).read()
for m in re.finditer(r'\n(_Z\w+): ; @', s):
    name = m.group(1)
    if not all(k in name for k in sys.argv[2:]):
        continue
    i = m.end(); j = s.find('.Lfunc_end', i)
    body = s[i:j]
    cnt = lambda pat: len(re.findall(pat, body))
    # main loop = largest basic-block span between a label and a backward branch; just report totals
    print(name)
    print('  v_readlane %d  v_fma_f64 %d  v_mul_f64 %d  ds_read %d  ds_write %d  s_nop %d  s_waitcnt %d  VALU %d  SALU %d  global_load %d  global_store %d  scratch %d'
          % (cnt('v_readlane'), cnt('v_fma_f64'), cnt('v_mul_f64'), cnt(r'ds_read'), cnt('ds_write'), cnt('s_nop'), cnt('s_waitcnt'),
             cnt(r'\n\s+v_'), cnt(r'\n\s+s_'), cnt('global_load'), cnt('global_store'), cnt('scratch_')))
