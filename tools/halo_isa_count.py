import re
s = open('/root/repo/casapose_amd/csrc/build/conv_halo-hip-amdgcn-amd-amdhsa-gfx950.s').read()
for cfg in ('Li1ELi1ELi2E', 'Li1ELi1ELi12E', 'Li1ELi1ELi0E', 'Li1ELi2ELi0E'):
    name = [m for m in re.findall(r"^(_ZN\S*conv_halo_kernelI%sEEvNS_5HaloKE):" % cfg, s, re.M)][0]
    i = s.index(name + ':'); k = s.index('.Lfunc_end', i)
    body = s[i:k].split('\n')
    bars = [n for n, l in enumerate(body) if 's_barrier' in l]
    def count(a, b):
        c = {'v': 0, 's': 0, 'mem': 0, 'lds': 0, 'br': 0}
        for l in body[a:b]:
            t = l.strip()
            if not t or t[0] in ';.': continue
            op = t.split()[0]
            if op.startswith('buffer_') or op.startswith('global_'): c['mem'] += 1
            elif op.startswith('ds_'): c['lds'] += 1
            elif op.startswith('s_cbranch') or op.startswith('s_branch'): c['br'] += 1
            elif op.startswith('v_'): c['v'] += 1
            elif op.startswith('s_'): c['s'] += 1
        return c
    print(cfg, 'producer steps:', [count(a, b) for a, b in zip(bars[-4:-1], bars[-3:])][1], ' consumer tap:', count(bars[2], bars[3]))
