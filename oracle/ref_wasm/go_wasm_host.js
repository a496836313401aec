// ORACLE TOOLING (test infrastructure, not product code).
//
// Minimal host for a Go 1.12 `GOOS=js GOARCH=wasm` module under node >= 12, written against
// the ABI table in SURVEY.md App. A.  It exists for one purpose: to execute the reference's
// OWN compiled prover, /root/reference/wasm/go-snark.wasm (built from
// wasm/go-snark-wasm-wrapper.go:21-26 -> generateProofs / verifyProofs / grothGenerateProofs /
// grothVerifyProofs), so that golden vectors under tests/golden/ come from the reference
// itself.  /root/reference only exists in the build container, so this script is run there
// (see gen_golden.js) and only its OUTPUT is committed.
//
// crypto/rand inside the wasm is `global.crypto.getRandomValues`; we replace it with a
// caller-supplied byte stream so Groth16's r and s (groth16/groth16.go:231-238 ->
// fields/fq.go:116-132, two 30-byte reads) are pinned.
"use strict";
const fs = require("fs");
const util = require("util");
const nodeCrypto = require("crypto");
const { performance } = require("perf_hooks");

const enc = new util.TextEncoder();
const dec = new util.TextDecoder("utf-8");

// ---- deterministic randomness -----------------------------------------------------------
let randStream = null;      // function(i) -> byte, or null for real randomness
let randPos = 0;
function setRandStream(fn) { randStream = fn; randPos = 0; }

function installGlobals() {
  global.fs = fs;
  global.performance = performance;
  global.crypto = {
    getRandomValues(buf) {
      if (randStream === null) { nodeCrypto.randomFillSync(buf); return; }
      for (let i = 0; i < buf.length; i++) buf[i] = randStream(randPos++) & 0xff;
    },
  };
}

class GoHost {
  constructor(opts) {
    this.quiet = !opts || opts.quiet !== false;
    this.exited = false;
    this._pendingEvent = null;
    this.timers = new Map();
    this.nextTimer = 1;
    const self = this;
    const view = () => new DataView(self.inst.exports.mem.buffer);
    const rdI64 = (a) => view().getUint32(a, true) + view().getInt32(a + 4, true) * 4294967296;
    const wrI64 = (a, v) => { view().setUint32(a, v >>> 0, true); view().setUint32(a + 4, Math.floor(v / 4294967296), true); };
    const NAN_HEAD = 0x7ff80000;

    const getVal = (a) => {
      const f = view().getFloat64(a, true);
      if (f === 0) return undefined;
      if (!isNaN(f)) return f;
      return self.values[view().getUint32(a, true)];
    };
    const boxed = (a, id, flag) => { view().setUint32(a + 4, NAN_HEAD | flag, true); view().setUint32(a, id, true); };
    const putVal = (a, v) => {
      if (typeof v === "number") {
        if (isNaN(v)) return boxed(a, 0, 0);
        if (v === 0) return boxed(a, 1, 0);
        return view().setFloat64(a, v, true);
      }
      if (v === undefined) return view().setFloat64(a, 0, true);
      if (v === null) return boxed(a, 2, 0);
      if (v === true) return boxed(a, 3, 0);
      if (v === false) return boxed(a, 4, 0);
      let id = self.refs.get(v);
      if (id === undefined) { id = self.values.length; self.values.push(v); self.refs.set(v, id); }
      const t = typeof v;
      boxed(a, id, t === "string" ? 1 : t === "symbol" ? 2 : t === "function" ? 3 : 0);
    };
    const bytesAt = (a) => new Uint8Array(self.inst.exports.mem.buffer, rdI64(a), rdI64(a + 8));
    const strAt = (a) => dec.decode(new DataView(self.inst.exports.mem.buffer, rdI64(a), rdI64(a + 8)));
    const valsAt = (a) => {
      const p = rdI64(a), n = rdI64(a + 8), out = new Array(n);
      for (let i = 0; i < n; i++) out[i] = getVal(p + 8 * i);
      return out;
    };
    const t0 = Date.now() - performance.now();

    this.importObject = { go: {
      "runtime.wasmExit": (sp) => { self.exitCode = view().getInt32(sp + 8, true); self.exited = true; },
      "runtime.wasmWrite": (sp) => {
        if (self.quiet) return;
        fs.writeSync(rdI64(sp + 8), new Uint8Array(self.inst.exports.mem.buffer, rdI64(sp + 16), view().getInt32(sp + 24, true)));
      },
      "runtime.nanotime": (sp) => wrI64(sp + 8, (t0 + performance.now()) * 1e6),
      "runtime.walltime": (sp) => { const ms = Date.now(); wrI64(sp + 8, ms / 1000); view().setInt32(sp + 16, (ms % 1000) * 1e6, true); },
      "runtime.scheduleTimeoutEvent": (sp) => {
        const id = self.nextTimer++;
        self.timers.set(id, setTimeout(() => self._resume(), rdI64(sp + 8) + 1));
        view().setInt32(sp + 16, id, true);
      },
      "runtime.clearTimeoutEvent": (sp) => { const id = view().getInt32(sp + 8, true); clearTimeout(self.timers.get(id)); self.timers.delete(id); },
      "runtime.getRandomData": (sp) => nodeCrypto.randomFillSync(bytesAt(sp + 8)),   // runtime seeds only
      "syscall/js.stringVal": (sp) => putVal(sp + 24, strAt(sp + 8)),
      "syscall/js.valueGet": (sp) => { const r = Reflect.get(getVal(sp + 8), strAt(sp + 16)); sp = self.inst.exports.getsp(); putVal(sp + 32, r); },
      "syscall/js.valueSet": (sp) => Reflect.set(getVal(sp + 8), strAt(sp + 16), getVal(sp + 32)),
      "syscall/js.valueIndex": (sp) => putVal(sp + 24, Reflect.get(getVal(sp + 8), rdI64(sp + 16))),
      "syscall/js.valueSetIndex": (sp) => Reflect.set(getVal(sp + 8), rdI64(sp + 16), getVal(sp + 24)),
      "syscall/js.valueCall": (sp) => {
        try {
          const v = getVal(sp + 8), m = Reflect.get(v, strAt(sp + 16)), args = valsAt(sp + 32);
          const r = Reflect.apply(m, v, args);
          sp = self.inst.exports.getsp();
          putVal(sp + 56, r); view().setUint8(sp + 64, 1);
        } catch (e) { putVal(sp + 56, e); view().setUint8(sp + 64, 0); }
      },
      "syscall/js.valueNew": (sp) => {
        try {
          const r = Reflect.construct(getVal(sp + 8), valsAt(sp + 16));
          sp = self.inst.exports.getsp();
          putVal(sp + 40, r); view().setUint8(sp + 48, 1);
        } catch (e) { putVal(sp + 40, e); view().setUint8(sp + 48, 0); }
      },
      "syscall/js.valueLength": (sp) => wrI64(sp + 16, parseInt(getVal(sp + 8).length)),
      "syscall/js.valuePrepareString": (sp) => { const s = enc.encode(String(getVal(sp + 8))); putVal(sp + 16, s); wrI64(sp + 24, s.length); },
      "syscall/js.valueLoadString": (sp) => bytesAt(sp + 16).set(getVal(sp + 8)),
      "debug": (v) => console.log(v),
    } };
  }

  start(instance) {
    this.inst = instance;
    this.values = [NaN, 0, null, true, false, global, instance.exports.mem, this];
    this.refs = new Map();
    // argv = ["js"], no environment: one C string at 4096, then the pointer table.
    const mem = new DataView(instance.exports.mem.buffer);
    let off = 4096;
    const arg = enc.encode("js\0");
    new Uint8Array(mem.buffer, off, arg.length).set(arg);
    const argPtr = off;
    off += 8;
    const argv = off;
    for (const p of [argPtr, 0, 0]) { mem.setUint32(off, p, true); mem.setUint32(off + 4, 0, true); off += 8; }
    instance.exports.run(1, argv);       // returns when main() blocks on its channel
  }

  _resume() {
    if (this.exited) throw new Error("go program exited");
    this.inst.exports.resume();
  }

  _makeFuncWrapper(id) {
    const host = this;
    return function () {
      const ev = { id: id, this: this, args: arguments };
      host._pendingEvent = ev;
      host._resume();
      return ev.result;
    };
  }
}

async function loadReferenceWasm(path, opts) {
  installGlobals();
  const host = new GoHost(opts);
  const { instance } = await WebAssembly.instantiate(fs.readFileSync(path), host.importObject);
  host.start(instance);
  await new Promise((r) => setTimeout(r, 100));
  for (const f of ["generateProofs", "verifyProofs", "grothGenerateProofs", "grothVerifyProofs"])
    if (typeof global[f] !== "function") throw new Error("reference wasm did not register " + f);
  return host;
}

module.exports = { loadReferenceWasm, setRandStream };
