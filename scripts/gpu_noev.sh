for i in 1 2; do
for f in "" "--event-stride 0"; do
python bench.py --steps 20 --warmup 2 --cpu-sample 0 --icp-pairs 0 --no-streamed --min-seconds 0.3 $f 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('%-20s value %8.0f fps  ms/step %.3f' % ('$f', d['value'], d['ms_per_step']))"
done; done
