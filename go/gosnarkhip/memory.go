package gosnarkhip

/*
#include "gosnark_hip.h"
*/
import "C"

// Device memory: accounting and eviction (include/gosnark_hip.h, "device memory").  A resident 2^20-constraint
// Groth16 key is 0.4 GiB of points plus 5.6 GiB of window tables built on its first proof; a server that keeps
// many keys resident releases the tables of the idle ones and pays ~140 ms on their next proof.
// Call sequence of Memory / HandleBytes / ReleaseTables / Trim = tests/c/memory_eviction.c.

// Memory mirrors gs_memory (bytes).
type Memory struct {
	DeviceTotal, DeviceFree uint64 // hipMemGetInfo of the GPU behind the logical device
	Library                 uint64 // every device byte the library holds, all logical devices
	Objects                 uint64 // handles of the logical device: keys, base arrays, scalars, R1CS
	Tables                  uint64 // their window tables
	Workspaces              uint64 // bucket sets, chunk partials, staging
	Handles                 uint64 // live handles
	Evictions               uint64 // window tables dropped because an allocation found no memory
}

// MemoryOf queries logical device `device`.
func MemoryOf(device int) (Memory, error) {
	var m C.gs_memory
	err := onDevice(device, func() C.int { return C.gs_memory_query(&m) })
	return Memory{uint64(m.device_total_bytes), uint64(m.device_free_bytes), uint64(m.library_bytes), uint64(m.object_bytes),
		uint64(m.table_bytes), uint64(m.workspace_bytes), uint64(m.objects), uint64(m.evictions)}, err
}

// HandleBytes returns what one handle holds: its own data and its window tables.
func HandleBytes(h Handle) (object, tables uint64, err error) {
	var a, b C.uint64_t
	err = call(func() C.int { return C.gs_handle_bytes(C.gs_handle(h), &a, &b) })
	return uint64(a), uint64(b), err
}

// ReleaseTables frees the window tables of a key or base array; outstanding tickets that read them finish
// first, and the handle's next proof / MSM rebuilds them.  Results never change.
func ReleaseTables(h Handle) error { return call(func() C.int { return C.gs_release_tables(C.gs_handle(h)) }) }

// ReleaseTables on a key.
func (k *Groth16Key) ReleaseTables() error   { return ReleaseTables(k.h) }
func (k *PinocchioKey) ReleaseTables() error { return ReleaseTables(k.h) }

// Trim frees every cached workspace of logical device `device` (rebuilt on demand).
func Trim(device int) error { return onDevice(device, func() C.int { return C.gs_trim() }) }
