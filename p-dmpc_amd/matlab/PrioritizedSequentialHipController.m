classdef PrioritizedSequentialHipController < PrioritizedSequentialController
    % PRIORITIZEDSEQUENTIALHIPCONTROLLER  The prioritized sequential controller with ALL computation levels of a time step
    % planned by ONE call into libpdmpc_hip.so (pdmpc_mex('plan_step', ...) -> pdmpc_plan_step: one kernel launch per step,
    % the predecessors' solved areas handed over on the device).
    %
    % It replaces exactly one method of PrioritizedSequentialController: controller()
    % (hlc/controller/prioritized/PrioritizedSequentialController.m:77-94), i.e. the double loop
    %
    %     for i_level = 1:max(levels_of_vehicles)
    %         for i_vehicle = find(levels_of_vehicles == i_level)
    %             obj.hlcs(i_vehicle).controller();   % plan() -> run_optimizer, publish_predictions()
    %
    % Everything around it (traffic info, coupling, prioritizing, grouping, fallbacks of coupled vehicles, apply) stays the
    % reference's.  Per vehicle the work of PrioritizedController.plan (PrioritizedController.m:297-354) is split:
    %   before the launch  the one-vehicle iter with the obstacles of its PARALLEL predecessors and its successors
    %                      (consider_parallel_coupling, consider_successors); the areas of SEQUENTIAL predecessors are not
    %                      read from messages -- the backend appends them on the device (PrioritizedController.m:476-491);
    %                      the areas the vehicle publishes should its search be exhausted (the fallback_areas argument):
    %                      handle_graph_search_exhaustion's standstill rectangle or plan_fallback's shifted previous plan
    %                      (:568-616, 678-718) -- what its sequential successors have to avoid in that case;
    %   after the launch   info from the record, handle_graph_search_exhaustion / plan_fallback for exhausted searches,
    %                      publish_predictions() in kahn order (the messages the next time step and the other groups read).
    %
    % Selected like the reference's controllers in HlcFactory (see INTEGRATION.md for the patch).  Shipped as source: no
    % MATLAB in the build environment of this backend; p-dmpc_amd/pdmpc/controller.py (step mode) and
    % csrc/step_controller.cpp are the tested twins of this logic, tests/test_matlab_marshal.py and
    % test_gpu_step.py::test_matlab_shaped_entry_points_plan_the_step_like_the_oracle test the call this class makes.

    properties (Access = public)
        n_gpus (1, 1) double = 1; % > 1: the step is planned over that many GPUs of this node (pdmpc_group_*, RCCL all-gather of the
        %                           solved areas: what PredictionsCommunication.send_message / read_message do between the vehicles'
        %                           processes, hlc/communication/PredictionsCommunication.m:34-63)
        shard_mode (1, 1) double = 0; % 0 auto (whole coupling-graph components, a dominating one by levels), 1 components, 2 levels
    end

    properties (Access = private)
        handle uint64 = uint64(0); % pdmpc_handle* (n_gpus == 1) or pdmpc_group* for all vehicles of the step
        mpa_uploaded (1, 1) logical = false;
        last_work double = []; % n_expanded of every vehicle's last plan: the weights of the next step's partition
    end

    methods

        function obj = PrioritizedSequentialHipController()
            obj@PrioritizedSequentialController();
        end

        function delete(obj)

            if obj.handle ~= 0 && obj.n_gpus > 1
                pdmpc_mex('group_destroy', obj.handle);
            elseif obj.handle ~= 0
                pdmpc_mex('destroy', obj.handle);
            end

        end

    end

    methods (Access = protected)

        function controller(obj)
            n = length(obj.hlcs);
            options = obj.hlcs(1).options;
            Hp = options.Hp;

            if obj.handle == 0
                checker = double(options.are_any_obstacles_non_convex); % OptimizerInterface.m:36-46

                if obj.n_gpus > 1
                    obj.handle = pdmpc_mex('group_create', Hp, checker, options.dt_seconds, n, obj.n_gpus);
                else
                    obj.handle = pdmpc_mex('create', Hp, checker, options.dt_seconds, n);
                end

            end

            if ~obj.mpa_uploaded
                mpa = obj.hlcs(1).mpa;

                if obj.n_gpus > 1
                    pdmpc_mex('group_upload_mpa', obj.handle, mpa.transition_matrix_single, mpa.maneuvers);
                else
                    pdmpc_mex('upload_mpa', obj.handle, mpa.transition_matrix_single, mpa.maneuvers);
                end

                obj.mpa_uploaded = true;
            end

            directed_coupling_sequential = obj.merged_graph("directed_coupling_sequential");
            levels_of_vehicles = kahn(directed_coupling_sequential);

            iters(1, n) = struct('x0', [], 'trim_index', [], 'reference_trajectory_points', [], 'v_ref', [], ...
                'obstacles', [], 'dynamic_obstacle_area', [], 'lanelet_boundary', [], 'hdv_reachable_sets', []);
            iter_v_all = cell(1, n);
            fallback_areas = cell(n, Hp);

            for i_vehicle = 1:n
                hlc = obj.hlcs(i_vehicle);
                [iter_v, fallback_row] = hlc.prepare_step_plan(); % see the companion patch of PrioritizedController below
                iter_v_all{i_vehicle} = iter_v;
                iters(i_vehicle) = pdmpc_iter_struct(iter_v);
                fallback_areas(i_vehicle, :) = fallback_row;
            end

            % ---- the whole double loop of PrioritizedSequentialController.controller: one call, one kernel launch
            if obj.n_gpus > 1
                % ... over the GPUs of the node: whole components of the coupling graph per GPU, the exchange of solved areas between
                % the levels of a shared component as an all-gather (pdmpc_group_plan_step); same records
                outs = pdmpc_mex('group_plan_step', obj.handle, iters, double(directed_coupling_sequential), fallback_areas, obj.last_work, obj.shard_mode);
                obj.last_work = double([outs.n_expanded]);
            else
                % (the work of every vehicle's last search as the expected work of this one: the launch hands its searches out by
                % priority, heavy ones and their predecessors first — pdmpc_set_step_weights; results are the same bit for bit)
                outs = pdmpc_mex('plan_step', obj.handle, iters, double(directed_coupling_sequential), fallback_areas, obj.last_work);
                obj.last_work = double([outs.n_popped]);
            end

            % ---- results, in kahn order (publish_predictions sends the messages later readers expect in this order)
            for i_level = 1:max(levels_of_vehicles)

                for i_vehicle = find(levels_of_vehicles == i_level)
                    obj.hlcs(i_vehicle).finish_step_plan(iter_v_all{i_vehicle}, outs(i_vehicle));
                end

            end

        end

    end

end

% -----------------------------------------------------------------------------------------------------------------
% Companion patch of hlc/controller/prioritized/PrioritizedController.m (two public methods next to plan(); they are
% plan() cut at the run_optimizer call, nothing else changes):
%
%   function [iter_v, fallback_row] = prepare_step_plan(obj)
%       obj.info = ControlResultsInfo(1, obj.options.Hp);
%       vehicle_index = obj.plant.vehicle_indices_controlled;
%       filter_self = false(1, obj.options.amount);
%       filter_self(vehicle_index) = true;
%       iter_v = IterationData.filter(obj.iter, filter_self);                                   % :300-304
%       predecessors = find(iter_v.directed_coupling(:, vehicle_index) == 1)';                  % :307
%       predecessors_sequential = find(iter_v.directed_coupling_sequential(:, vehicle_index))'; % :309
%       successors = find(iter_v.directed_coupling(vehicle_index, :) == 1);                     % :311
%       % parallel predecessors only: the sequential ones are handed over on the device         % :494-503
%       predecessors_parallel = setdiff(predecessors, predecessors_sequential);
%       dynamic_obstacle_area_predecessors = cell(0, obj.options.Hp);
%       for j_vehicle = predecessors_parallel
%           dynamic_obstacle_area_predecessors = [dynamic_obstacle_area_predecessors; obj.consider_parallel_coupling(j_vehicle)]; %#ok<AGROW>
%       end
%       [obstacles_successors, dynamic_obstacle_area_successors] = obj.consider_successors(successors);      % :321
%       iter_v.obstacles = [iter_v.obstacles; obstacles_successors];                                          % :324
%       iter_v.dynamic_obstacle_area = [iter_v.dynamic_obstacle_area; dynamic_obstacle_area_predecessors; dynamic_obstacle_area_successors];
%       % what this vehicle publishes if its search is exhausted (:568-616 standstill, :678-718 previous plan shifted)
%       if obj.mpa.trims(iter_v.trim_indices).speed == 0 && ~(obj.options.constraint_from_successor == ConstraintFromSuccessor.none)
%           vehiclePolygon = transformed_rectangle(iter_v.x0(1, 1), iter_v.x0(1, 2), iter_v.x0(1, 3), ...
%               obj.scenario_adapter.scenario.vehicles(1).Length, obj.scenario_adapter.scenario.vehicles(1).Width);
%           fallback_row = repmat({[vehiclePolygon, vehiclePolygon(:, 1)]}, 1, obj.options.Hp);
%       elseif obj.k > 1
%           fallback_row = del_first_rpt_last(obj.info_old.shapes);
%       else
%           fallback_row = cell(1, obj.options.Hp);
%       end
%   end
%
%   function finish_step_plan(obj, iter_v, out)
%       obj.timing.start('plan', obj.k);
%       obj.info = GraphSearchHip.info_from_record(iter_v, obj.options, out);   % the tail of GraphSearchHip.run_optimizer
%       if obj.info.is_exhausted                                                % :344-352
%           obj.info = obj.handle_graph_search_exhaustion(obj.info, iter_v);
%           if obj.info.needs_fallback
%               obj.plan_fallback();
%           end
%       end
%       obj.timing.stop('plan', obj.k);
%       obj.publish_predictions();                                              % :291-293
%   end
