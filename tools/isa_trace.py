import re, sys
s = open('/root/repo/casapose_amd/csrc/build/conv_f32-hip-amdgcn-amd-amdhsa-gfx950.s').read()
cfg = sys.argv[1] if len(sys.argv) > 1 else '2,2,2,2,0'
name = '_ZN12_GLOBAL__N_115conv_f32_kernelI' + ''.join('Li%sE' % x for x in cfg.split(',')) + 'EEvNS_5ConvKE'
i = s.index(name + ':'); j = s.index('s_endpgm', i)
body = s[i:j].split('\n')
idx = [n for n, l in enumerate(body) if 'v_mfma' in l]
print('mfma', len(idx), idx[0], idx[-1], 'lines', len(body))
seq = []
for n in range(max(0, idx[0] - 150), idx[-1] + 30):
    t = body[n].strip()
    if not t or t.startswith(';') or t.startswith('.'):
        if t.startswith('.LBB'): seq.append('\n' + t.split(':')[0] + ':')
        continue
    op = t.split()[0]
    if op.startswith('v_mfma'): c = 'M'
    elif op.startswith('ds_read'): c = 'r'
    elif op.startswith('ds_write'): c = 'w'
    elif op.startswith('global_load') or op.startswith('buffer_load'): c = 'G'
    elif op.startswith('s_waitcnt'): c = '[' + t.split(None, 1)[1].replace(' ', '') + ']'
    elif op.startswith('s_barrier'): c = '|BAR|'
    elif op.startswith('s_cbranch') or op.startswith('s_branch'): c = '{br}'
    elif op.startswith('v_'): c = 'v'
    elif op.startswith('s_'): c = 's'
    else: c = '?'
    seq.append(c)
print(''.join(seq))
t = open('/root/repo/casapose_amd/csrc/build/conv_remarks.txt').read()
for n, v, sc, occ, sp in re.findall(r"Function Name: (\S+).*?VGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+).*?VGPRs Spill: (\d+)", t, re.S):
    if int(sp) or int(sc): print('SPILL', n, v, sc, sp)
    if name in n: print('vgpr', v, 'occ', occ)
