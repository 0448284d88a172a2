import json,sys
d=json.load(open(sys.argv[1]))
ops=d if isinstance(d,list) else d.get('ops',d)
tot=0
for o in ops:
    ms=o.get('ms',0); tot+=ms
    print(f"{o.get('name','?'):40s} {o.get('kernel','')[:40]:40s} {ms*1000:8.1f}")
print('total',tot)
