// The cgo side of libgosnark_hip.so: gosnarkhip (the binding) and the drop-in packages groth16hip, snarkhip, r1csqaphip,
// bn128hip, which keep the signatures of github.com/arnaucube/go-snark-study's groth16, root (snark), r1csqap and bn128 packages.
//
// Building needs: a Go toolchain with cgo (CGO_ENABLED=1), a checkout of the reference beside this repository (the
// `replace` below; point it at the module cache instead by deleting it), libgosnark_hip.so built by
// `make -C go-snark-study_amd/csrc` (the #cgo LDFLAGS of gosnarkhip/gosnarkhip.go link and rpath it) and an MI355X at run time.
//   cd go && go vet ./... && go build ./...
// The build image of this repository has no Go toolchain: these packages are reviewed, not compiled, there; every exported
// function's C call sequence is a compiled-and-run C program under tests/c/ (INTEGRATION.md lists the pairs).
module github.com/arnaucube/go-snark-study-hip

go 1.12

require github.com/arnaucube/go-snark-study v0.0.0-00010101000000-000000000000

replace github.com/arnaucube/go-snark-study => ../../go-snark-study
