// The wide hierarchy built on the device (bvh_device.hip), as a drop-in for the host's SAH
// build + collapse inside flatten_scene.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/mi355rt.h"

namespace rt {

// leaf_boxes: one record per leaf slot, float boxes already rounded outwards (a, b unused).
// records: the hierarchy in the host builder's format (breadth-first, children consecutive,
// leaf records naming one position of leaf_order); leaf_order: leaf slots in leaf order;
// depth: levels below the root.  Returns "" or an error message.  The current device is used.
std::string device_build_wide(hipStream_t stream, const std::vector<rt_bvh_node>& leaf_boxes,
                              std::vector<rt_bvh_node>& records, std::vector<uint32_t>& leaf_order,
                              uint32_t& depth);

// The objects in the order the reference prefers among hits at exactly the same t
// (BvhNode::new_along_axis, src/hittable/bvh.rs:33-77 with :93-97 and hittable_list.rs:31): what
// the host's reference_order computes, by sorts on the device.  centres: the centre of every
// object's reference box along x, then y, then z (n each).
std::string device_reference_order(hipStream_t stream, const std::vector<double>& centres, uint32_t n,
                                   std::vector<uint32_t>& order);

}  // namespace rt
