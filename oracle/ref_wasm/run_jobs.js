// ORACLE TOOLING (test infrastructure, not product code).
// Runs a list of prove/verify jobs through the reference's compiled wasm prover and writes
// the raw results.  Usage:  node run_jobs.js <go-snark.wasm> <jobs.json> <out.json>
// jobs.json: [{name, kind: "groth"|"pinocchio", circuit: <json text>, setup: <json text>,
//              px: <json text>, inputs: <json text>, rand: [bytes...] | null,
//              verify: [<public inputs json text>, ...]}]
// All four payloads are passed through as TEXT (JSON.stringify would destroy big integers).
// Special kind "fixture": extracts the Pinocchio demo fixture from wasm/index.js (:6-8).
"use strict";
const fs = require("fs");
const path = require("path");
const { loadReferenceWasm, setRandStream } = require("./go_wasm_host.js");

function indexJsFixture(wasmPath) {
  // wasm/index.js:6-8 hold `const circuit = {...}`, `const setup = {...}`, `const px = [...]`
  const src = fs.readFileSync(path.join(path.dirname(wasmPath), "index.js"), "utf8").split("\n");
  const grab = (name) => {
    const line = src.find((l) => l.startsWith("const " + name + " = "));
    return line.slice(("const " + name + " = ").length).replace(/;\s*$/, "");
  };
  return { circuit: grab("circuit"), setup: grab("setup"), px: grab("px") };
}

(async () => {
  const [wasmPath, jobsPath, outPath] = process.argv.slice(2);
  await loadReferenceWasm(wasmPath, { quiet: true });
  const jobs = JSON.parse(fs.readFileSync(jobsPath, "utf8"));
  const out = [];
  for (const job of jobs) {
    let { circuit, setup, px, inputs } = job;
    if (job.kind === "fixture") {
      const fx = indexJsFixture(wasmPath);
      circuit = fx.circuit; px = fx.px;
      // The fixture predates the current SetupString layout (utils/base10parsers.go:135-147):
      // G1T sits at the top level; the prover reads Pk.G1T (snark.go:285).
      const s = JSON.parse(fx.setup);      // all numbers are strings here, safe to parse
      s.Pk.G1T = s.G1T;
      setup = JSON.stringify(s);
    }
    const rec = { name: job.name, kind: job.kind, circuit, setup, px, inputs, rand: job.rand || null };
    if (job.rand) setRandStream((i) => job.rand[i]); else setRandStream(null);
    const t0 = Date.now();
    const prove = (job.kind === "groth") ? global.grothGenerateProofs : global.generateProofs;
    rec.proof = prove(circuit, setup, px, inputs);
    rec.prove_ms = Date.now() - t0;
    rec.verify = [];
    for (const pub of job.verify || []) {
      const v = (job.kind === "groth") ? global.grothVerifyProofs : global.verifyProofs;
      rec.verify.push({ public: pub, result: v(setup, rec.proof, pub) });
    }
    out.push(rec);
    process.stderr.write(`[ref-wasm] ${job.name}: prove ${rec.prove_ms} ms, verify ${JSON.stringify(rec.verify.map((x) => x.result))}\n`);
  }
  fs.writeFileSync(outPath, JSON.stringify(out));
  process.exit(0);
})().catch((e) => { console.error(e); process.exit(1); });
