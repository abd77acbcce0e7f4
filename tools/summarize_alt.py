import re, sys, collections
rows = collections.OrderedDict()
for line in open(sys.argv[1]):
    m = re.match(r'(.{20}) waves (\d) pad\s+(\d+) overlap (\d) B(\d) alt(\d): step\s+([\d.]+) ms.*A sum\s+([\d.]+) ms\s+B sum\s+([\d.]+) ms\s+same=(\w+)', line)
    if m:
        key = (m.group(1).strip(), m.group(2), m.group(3), m.group(4), m.group(5))
        rows.setdefault(key, {}).setdefault(int(m.group(6)), []).append((float(m.group(7)), float(m.group(8)), m.group(10)))
for key, v in rows.items():
    if 0 in v and 1 in v:
        s0 = min(x[0] for x in v[0]); s1 = min(x[0] for x in v[1]); a0 = min(x[1] for x in v[0]); a1 = min(x[1] for x in v[1])
        ok = all(x[2] == 'True' for x in v[0] + v[1])
        print(f'{key[0]:20s} w{key[1]} ov{key[3]}: step {s0:7.3f} -> {s1:7.3f} ({100 * (s0 / s1 - 1):+5.1f} %)  A sum {a0:7.3f} -> {a1:7.3f} ({100 * (a0 / a1 - 1):+5.1f} %)  same={ok}')
