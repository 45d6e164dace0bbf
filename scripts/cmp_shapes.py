import json,sys
a=json.load(open(sys.argv[1])); b=json.load(open(sys.argv[2]))
sa=a['shapes']; sb=b['shapes']
def keyed(s,op):
    return {r['shape'].replace(op+' ',''):r for r in s if r['shape'].startswith(op+' ')}
ka=keyed(sa,'conv3x3_hp'); kb=keyed(sb,'conv3x3_bf3')
tot_a=tot_b=0
for k in sorted(kb, key=lambda k:-kb[k]['total_ms']):
    if k in ka:
        ra,rb=ka[k],kb[k]
        tot_a+=ra['total_ms']; tot_b+=rb['total_ms']
        print("%-58s n=%3d  bf3 %7.1f us  hp %7.1f us  %+5.1f%%" % (k[:58], rb['launches'], rb['avg_us'], ra['avg_us'], (ra['avg_us']/rb['avg_us']-1)*100))
print("total ms: hp %.2f  bf3 %.2f" % (tot_a, tot_b))
