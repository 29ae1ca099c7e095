classdef GraphSearchHip < OptimizerInterface
    % GRAPHSEARCHHIP  Optimal graph search on an AMD MI355X through libpdmpc_hip.so.
    %
    % Drop-in replacement for GraphSearch (hlc/optimizer/graph_search/GraphSearch.m): same
    % run_optimizer signature and the same ControlResultsInfo fields.  Selected with
    % options.optimizer_type = OptimizerType.HipOptimal (see INTEGRATION.md for the three-line
    % patch to OptimizerInterface.get_optimizer and OptimizerType).
    %
    % This file is shipped as source; it cannot be executed in the build environment of this
    % backend (no MATLAB there).  The Python class pdmpc.optimizer.GraphSearchHip is the tested
    % twin of this class.

    properties (Access = private)
        handle uint64 = uint64(0); % pdmpc_handle*, owned by the MEX file
        mpa_uploaded (1, 1) logical = false;
    end

    methods

        function obj = GraphSearchHip(options)
            obj = obj@OptimizerInterface();
            % checker follows OptimizerInterface.set_constraint_checker (OptimizerInterface.m:36-46)
            checker = double(options.are_any_obstacles_non_convex); % 0 = SAT, 1 = InterX
            obj.handle = pdmpc_mex('create', options.Hp, checker, options.dt_seconds);
        end

        function delete(obj)

            if obj.handle ~= 0
                pdmpc_mex('destroy', obj.handle);
            end

        end

        function info = run_optimizer(obj, ~, iter, mpa, options, ~)
            % Same contract as GraphSearch.run_optimizer (GraphSearch.m:14-17): one vehicle.
            assert(iter.amount == 1);

            if ~obj.mpa_uploaded
                pdmpc_mex('upload_mpa', obj.handle, mpa.transition_matrix_single, mpa.maneuvers);
                obj.mpa_uploaded = true;
            end

            % libpdmpc_hip.so flattens the cells of iter into pdmpc_vehicle_in (csrc/matlab_marshal.cpp) and returns pdmpc_vehicle_out
            out = pdmpc_mex('plan', obj.handle, pdmpc_iter_struct(iter));

            info = GraphSearchHip.info_from_record(iter, options, out);
        end

    end

    methods (Static)

        function info = info_from_record(iter, options, out)
            % pdmpc_vehicle_out (as the struct pdmpc_mex returns) -> ControlResultsInfo; also used by
            % PrioritizedSequentialHipController, which gets one record per vehicle from a single call
            info = ControlResultsInfo(iter.amount, options.Hp);
            info.n_expanded = out.n_expanded;
            % Only an empty open list is an exhaustion (GraphSearch.m:57-61).  pdmpc_plan_batch re-plans with doubled arenas
            % when a search outgrows its arena, so PDMPC_ARENA_OVERFLOW (2) only comes back when HBM (or the limit set with
            % pdmpc_set_arena_limit) is used up; the reference's tree is unbounded (Tree.m:54-70) and would have kept searching,
            % so that is an error of this backend, not a planner fallback.  Negative values are device-side errors.
            if out.status == 2
                error('GraphSearchHip:arena', 'search tree outgrew the arena and could not be grown (status PDMPC_ARENA_OVERFLOW)');
            elseif out.status < 0
                error('GraphSearchHip:backend', 'pdmpc_plan_batch reported status %d in the result record', out.status);
            end
            info.is_exhausted = out.status == 1; % PDMPC_EXHAUSTED

            if info.is_exhausted
                return % y_predicted stays NaN (ControlResultsInfo.m:40); the caller decides on the fallback
            end

            Hp = options.Hp;
            % rows in NodeInfo order (NodeInfo.m:5-13), as create_control_results_info_from_mex expects
            next_nodes = arrayfun(@(k) out.path_nodes(k + 1, :), 1:Hp, UniformOutput = false);
            trims = [iter.trim_indices, out.predicted_trims(1:Hp)];
            y_full = {[out.y_predicted(1:Hp, :), zeros(Hp, 1)]}; % one row per step: entries_per_time_step == 1
            info = OptimizerInterface.create_control_results_info_from_mex(info, iter, options, next_nodes, trims, y_full);
            % the helper does not fill shapes / needs_fallback; publish_predictions reads info.shapes(1, :)
            info.shapes = arrayfun(@(k) squeeze(out.shapes(k, :, 1:out.shape_cols(k))), 1:Hp, UniformOutput = false);
            info.tree_path = 1:(Hp + 1);
            info.needs_fallback = false;
        end

    end

end
