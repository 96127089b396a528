import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1])
o=[k for k in ("roofline_encode","roofline_decode") if k in b][0]
enc = b["roofline"] if "encode" in b["roofline"]["kernel"] else b[o]
dec = b["roofline"] if "decode" in b["roofline"]["kernel"] else b[o]
pm=b["peak_measured"]
print(f"HEAD value {b['value']/1e6:.3f} M  step {b['ms_per_step']:.4f} ms | encode kernel {enc['avg_launch_ms']:.4f} ({enc['frac']:.3f})  decode kernel behind it {dec['avg_launch_ms']:.4f} ({dec['frac']:.3f})  dominant: {b['roofline']['kernel'][:15]} | encode/call {b['encode_ms']:.4f} decode-only {b['decode_ms']:.4f} | read {pm['read_GBps']:.0f} write {pm['write_GBps']:.0f} copy {pm['copy_GBps']:.0f} misaligned {pm['write_misaligned_GBps']:.0f}")
